"""Query-stage splitting of a physical plan, on its serde_json form -- the host-side step that turns one plan into the
stage plans the distributed mode ships to its functions.

`build_query_dag` restates `flock/src/distributed_plan/stage.rs:269-367` (`build_query_dag_from_serde_json`, reached
through `DistributedPlanner::plan_query_stages`, flock/src/distributed_plan/planner.rs:48-63): walk down the `input` chain
from the root;
  * `hash_aggregate_exec` in mode Final / FinalPartitioned: everything above and including it becomes a stage whose new
    leaf is an empty `memory_exec` with the schema of the cut-off input; the walk continues on the cut-off input;
  * `hash_join_exec`: the current plan (join with two empty `memory_exec` leaves) becomes a stage, its `left` and `right`
    inputs become the two plans of the stage below -- and the walk stops there (stage.rs:335-336);
  * `sort_exec`: like a final aggregate;
  * everything else: descend.
A stage plan whose root is `coalesce_batches_exec <- repartition_exec Hash` is a shuffling stage
(`ExecutionContext::is_shuffling`, flock/src/runtime/context.rs:328-337): the function runs `execute_partitioned` and sends
partition j to member j of the next function group (flock-function/src/aws/actor.rs:60-66,425-543).

`split_at_repartitions` is the finer rule of the playground's ShuffleWriter plans
(playground/src/distributed_plan/nexmark/q{3,5,8}.dag): a cut at EVERY hash repartition and at every
CoalescePartitions (gather), so that no stage repartitions in its middle and every stage can run once per partition.

Both return a list of `Stage`s, leaves first; `Stage.inputs[i]` names the stage that feeds the i-th `memory_exec` leaf
(in the plan's leaf order, the order `feed_data_sources` walks: breadth-first), or None for a leaf fed by a base relation.
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import List, Optional


@dataclass
class Stage:
    plan: dict
    inputs: List[Optional[int]] = field(default_factory=list)   # per memory_exec leaf (BFS order): producing stage or None
    # node of the reference's QueryDag this plan belongs to: a join's left and right sub-plans are ONE node holding two plans
    # (`dag.insert(leaf, vec![left, right], ..)`, stage.rs:330-334 -- one function whose CloudExecutionPlan carries both), every other
    # stage plan is a node of its own.  `dag_nodes(stages)` groups the stages that way; node and edge counts then equal the ones
    # stage.rs's own tests assert (3 nodes / 2 edges for sort + limit over a join, stage.rs:776-901).
    node: int = -1

    def plan_has_join(self) -> bool:
        n = self.plan
        while isinstance(n, dict):
            if n.get("execution_plan") == "hash_join_exec":
                return True
            n = n.get("input")
        return False

    @property
    def is_shuffling(self) -> bool:
        t = self.plan
        return (t.get("execution_plan") == "coalesce_batches_exec" and isinstance(t.get("input"), dict)
                and t["input"].get("execution_plan") == "repartition_exec" and "Hash" in t["input"].get("partitioning", {}))


def node_schema(n: dict) -> dict:
    """Output schema of a plan node (the `schema()` the splitter asks DataFusion for, stage.rs:289,321-324)."""
    kind = n.get("execution_plan")
    if kind == "memory_exec":
        fields = n["schema"]["fields"]
        proj = n.get("projection")
        if proj and all(isinstance(i, int) and 0 <= i < len(fields) for i in proj):
            fields = [fields[i] for i in proj]
        return {"fields": copy.deepcopy(fields), "metadata": {}}
    if kind in ("projection_exec", "hash_aggregate_exec", "hash_join_exec") and "schema" in n:
        return copy.deepcopy(n["schema"])
    if "input" in n:
        return node_schema(n["input"])
    raise ValueError(f"no schema for plan node {kind!r}")


def _empty_memory(schema: dict) -> dict:
    """`MemoryExec::try_new(&[], schema, None)` serialised (stage.rs:287-291)."""
    return {"execution_plan": "memory_exec", "schema": schema, "projection": None}


def leaves_bfs(plan: dict) -> List[dict]:
    """memory_exec leaves in the breadth-first order feed_data_sources visits them (context.rs:262-300)."""
    out, queue = [], [plan]
    while queue:
        n = queue.pop(0)
        kids = [n[k] for k in ("input", "left", "right") if isinstance(n.get(k), dict)]
        if n.get("execution_plan") == "memory_exec":
            out.append(n)
        queue.extend(kids)
    return out


def build_query_dag(plan: dict) -> List[Stage]:
    """stage.rs:269-367 on the JSON tree.  Returns stages leaves-first; the last one is the plan's root stage."""
    root = copy.deepcopy(plan)
    stages_top_down: List[tuple] = []     # (plan, marker leaf objects in it that the NEXT entries feed)
    node = root
    while True:
        kind = node.get("execution_plan")
        if kind == "hash_aggregate_exec" and node.get("mode") in ("Final", "FinalPartitioned") or kind == "sort_exec":
            below = node["input"]
            leaf = _empty_memory(node_schema(below))
            node["input"] = leaf
            stages_top_down.append((root, [leaf], 1))
            root = node = below
            continue
        if kind == "hash_aggregate_exec" and node.get("mode") != "Partial":
            raise ValueError("Failed to parse aggregate mode for HashAggregateExec")     # stage.rs:300-305
        if kind == "hash_join_exec":
            left, right = node["left"], node["right"]
            ll, rl = _empty_memory(node_schema(left)), _empty_memory(node_schema(right))
            node["left"], node["right"] = ll, rl
            stages_top_down.append((root, [ll, rl], 2))
            stages_top_down.append((left, [], 0))
            stages_top_down.append((right, [], 0))
            root = None
            break
        if not isinstance(node.get("input"), dict):
            break
        node = node["input"]
    if root is not None:
        stages_top_down.append((root, [], 0))
    # leaves first
    order = list(reversed(range(len(stages_top_down))))
    index_of = {top: i for i, top in enumerate(order)}
    stages: List[Stage] = []
    for top in order:
        plan_t, markers, n_children = stages_top_down[top]
        feeders = [index_of[top + 1 + k] for k in range(n_children)]
        ins = []
        for lf in leaves_bfs(plan_t):
            hit = [feeders[k] for k, m in enumerate(markers) if m is lf]
            ins.append(hit[0] if hit else None)
        stages.append(Stage(plan_t, ins))
    # reference DAG nodes: the two inputs of a join (consecutive, fed by base relations, consumed by the same stage) share one
    node = 0
    consumer = {}
    for i, st in enumerate(stages):
        for j in st.inputs:
            if j is not None:
                consumer[j] = i
    for i, st in enumerate(stages):
        sibling = i - 1
        if (i > 0 and i in consumer and consumer.get(sibling) == consumer[i] and stages[consumer[i]].plan_has_join()
                and not any(j is not None for j in st.inputs) and not any(j is not None for j in stages[sibling].inputs)):
            st.node = stages[sibling].node
        else:
            st.node = node
            node += 1
    return stages


def dag_nodes(stages: List[Stage]) -> List[List[int]]:
    """Stage indices grouped by reference QueryDag node (leaves first)."""
    out = {}
    for i, st in enumerate(stages):
        out.setdefault(st.node if st.node >= 0 else 10_000 + i, []).append(i)
    return [out[k] for k in sorted(out)]


def dag_edge_count(stages: List[Stage]) -> int:
    """Edges between reference DAG nodes (a join's two inputs reach their consumer through one edge)."""
    edges = set()
    for i, st in enumerate(stages):
        for j in st.inputs:
            if j is not None:
                edges.add((stages[j].node, st.node))
    return len(edges)


def split_at_repartitions(plan: dict) -> List[Stage]:
    """A cut at every `repartition_exec` Hash (the playground's ShuffleWriter stages, q3.dag / q5.dag / q8.dag): the
    sub-tree `coalesce_batches_exec? <- repartition_exec Hash <- X` becomes its own shuffling stage and is replaced by an
    empty memory_exec of its schema.  A Hash repartition directly under the plan's root stays where it is."""
    stages: List[Stage] = []

    def is_hash(n):
        return isinstance(n, dict) and n.get("execution_plan") == "repartition_exec" and "Hash" in n.get("partitioning", {})

    def is_gather(n):   # CoalescePartitionsExec / MergeExec: every partition of the input meets in ONE consumer
        return isinstance(n, dict) and n.get("execution_plan") in ("coalesce_partitions_exec", "merge_exec")

    def cut(n: dict, is_root: bool) -> dict:
        """Returns `n` with every shuffling sub-tree below it replaced by a marker leaf."""
        n = dict(n)
        for key in ("input", "left", "right"):
            child = n.get(key)
            if not isinstance(child, dict):
                continue
            sub = child
            wrapped = sub.get("execution_plan") == "coalesce_batches_exec" and is_hash(sub.get("input"))
            if is_gather(sub):
                # the input of a gather runs once per partition and is not shuffled again (q5.dag: `ShuffleWriterExec: None`
                # above the Partial MAX); the gather itself stays in the consumer, which then sees every partition's rows
                idx = emit(cut(sub["input"], True))
                leaf = _empty_memory(node_schema(sub["input"]))
                leaf["__stage__"] = idx
                n[key] = dict(sub, input=leaf)
            elif (wrapped or is_hash(sub)) and not (is_root and n.get("execution_plan") == "coalesce_batches_exec"):
                stage_plan = cut(sub, True) if wrapped else {"execution_plan": "coalesce_batches_exec", "input": cut(sub, True),
                                                              "target_batch_size": 4096}
                idx = emit(stage_plan)
                leaf = _empty_memory(node_schema(sub))
                leaf["__stage__"] = idx
                n[key] = leaf
            else:
                n[key] = cut(child, is_root and n.get("execution_plan") == "coalesce_batches_exec")
        return n

    def emit(stage_plan: dict) -> int:
        ins = []
        for lf in leaves_bfs(stage_plan):
            ins.append(lf.pop("__stage__", None))
        stages.append(Stage(stage_plan, ins))
        return len(stages) - 1

    emit(cut(copy.deepcopy(plan), True))
    return stages


def stage_levels(stages):
    """The reference numbers stages by depth ("=== Stage 0 ===" prints both base-fed plans of a join, planner.rs:148-171):
    level 0 = plans fed by base relations only, level k = 1 + the deepest producer.  Returns one level per stage plan."""
    levels = []
    for st in stages:
        feeders = [j for j in st.inputs if j is not None]
        levels.append(1 + max(levels[j] for j in feeders) if feeders else 0)
    return levels


class StagedRun:
    """The distributed run of a stage DAG in ONE process: every stage is a function group with its plan instantiated once
    (`ExecutionContext`), a shuffling stage is executed with `execute_partitioned` (through `collect`) by `chunks` producers over
    slices of its base relation, and partition j of every producer goes to invocation j of the consuming stage -- the routing of
    flock-function/src/aws/actor.rs:425-543 without the Lambda invocations in between.  What the reference's
    `launcher/aws` differential tests do with real functions (flock/src/launcher/aws/mod.rs:423-468)."""

    def __init__(self, gpu, stages: List[Stage], chunks: int = 1, instances: int = 0, share_sources: bool = False, on_device: bool = False):
        """instances: function instances of a consuming stage.  0 = one per hash partition (the reference's default: the stage's
        concurrency equals its input partitioning); k > 0 = partition p goes to instance p % k, which feeds everything it is sent
        into ONE execute -- a GPU function hosting several partitions.  The union over the instances is the same multiset either
        way: the partitions are disjoint in the key the stage joins / groups on (what `HashJoinExec mode=Partitioned` and
        `FinalPartitioned` rest on), so a join or an aggregate over several partitions at once is the union of the per-partition
        results.
        share_sources: the stages that read base relations are fed ONE device copy of a relation between them (chunks = 1;
        `ExecutionContext.share_data_sources`): q5's two subplans both scan `bid`.
        on_device: one instance per stage, and a stage's result stays in HBM for the stages that consume it
        (`flockgpu_plan_execute_retain` / `flockgpu_plan_feed_from`): only the base relations come from the host and only the last
        stage's result goes back to it."""
        from .runtime import ExecutionContext
        self.stages, self.chunks, self.instances, self.share_sources, self.on_device = stages, chunks, instances, share_sources, on_device
        self.ctxs = [ExecutionContext([st.plan], name=f"stage-{i}", gpu=gpu) for i, st in enumerate(stages)]

    def close(self):
        for c in self.ctxs:
            c.close()

    def run(self, relations):
        """relations: {name: RecordBatch} of the base relations (one window).  Returns the root stage's batches."""
        from .runtime import collect
        if self.on_device:
            return self._run_on_device(relations)
        outputs = {}
        base = [i for i, st in enumerate(self.stages) if any(j is None for j in st.inputs)]
        if self.share_sources and self.chunks == 1 and len(base) > 1:
            # one upload per relation: the first source stage is fed, the others read its device copy; all execute before any cleans
            src = [[[rb]] for rb in relations.values()]
            donor = self.ctxs[base[0]]
            donor.feed_data_sources(src)
            for i in base[1:]:
                if not self.ctxs[i].share_data_sources(donor):
                    self.ctxs[i].feed_data_sources(src)
            for i in base:
                ctx = self.ctxs[i]
                outputs[i] = [p for p in ctx.execute_partitioned()[0]] if self.stages[i].is_shuffling else [ctx.execute()[0]]
            for i in reversed(base):
                self.ctxs[i].clean_data_sources()
        for i, (st, ctx) in enumerate(zip(self.stages, self.ctxs)):
            if i in outputs:
                continue
            feeders = [j for j in st.inputs if j is not None]
            invocations = []
            if any(j is None for j in st.inputs):
                for c in range(self.chunks):
                    src = []
                    for rb in relations.values():
                        lo, hi = rb.num_rows * c // self.chunks, rb.num_rows * (c + 1) // self.chunks
                        src.append([[rb.slice(lo, hi - lo)]])
                    invocations.append(src)
            else:
                parts = max(len(outputs[j]) for j in feeders)
                k = min(self.instances, parts) if self.instances > 0 else parts
                for inst in range(k):
                    mine = range(inst, parts, k)
                    invocations.append([[[b for p in mine for b in outputs[j][p]] if len(outputs[j]) == parts
                                         else [b for part in outputs[j] for b in part]] for j in feeders])
            result = None
            for src in invocations:
                out = collect(ctx, src)
                if st.is_shuffling:
                    result = result or [[] for _ in out]
                    for p, batches in enumerate(out):
                        result[p].extend(batches)
                else:
                    result = result or [[]]
                    result[0].extend(out[0])
            outputs[i] = result
        return [b for part in outputs[len(self.stages) - 1] for b in part]

    def _run_on_device(self, relations):
        src = [[[rb]] for rb in relations.values()]
        last = len(self.stages) - 1
        donor, out = None, None
        try:
            for i, (st, ctx) in enumerate(zip(self.stages, self.ctxs)):
                if any(j is None for j in st.inputs):
                    if donor is None or not (self.share_sources and ctx.share_data_sources(donor)):
                        ctx.feed_data_sources(src)
                        donor = donor or ctx
                else:
                    ctx.feed_from([self.ctxs[j] for j in st.inputs if j is not None])
                if i == last:
                    out = ctx.execute()[0]
                else:
                    ctx.execute_retain()
        finally:
            for ctx in reversed(self.ctxs):   # consumers first: a producer's buffers are only released once nothing reads them
                ctx.clean_data_sources()
        return out
