/* A C99 consumer of the drop-in boundary (include/flockgpu.h, include/flockgpu_plan.h) -- no ctypes, no C++: what a Rust
 * `extern "C"` block binds is exactly what this file calls.  It restates one `actor::collect`
 * (flock-function/src/aws/actor.rs:54-79): plan JSON in, feed two hand-built Arrow batches, execute, walk the returned
 * ArrowArray, release it, reset, run a second invocation; then the round-4 surface (pane ring, asynchronous execute + wait, the
 * partition-scheme check, guarded memory); destroy.
 *
 *   consumer <plan.json> <modulus>      (the plan is NEXMark q2: Filter auction % modulus = 0 -> [auction, price])
 *
 * prints one line per result row "auction price" per invocation, "--" between invocations, and exits non-zero with the library's
 * message on any failure.  Built by tests/test_c_consumer.py with `gcc -std=c99 -pedantic -Wall -Werror -Iinclude`. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "flockgpu.h"
#include "flockgpu_plan.h"

static void noop_release_array(struct ArrowArray *a) { a->release = NULL; }
static void noop_release_schema(struct ArrowSchema *s) { s->release = NULL; }

/* a struct<auction:int32, price:int32> batch over caller-owned buffers */
typedef struct {
    struct ArrowArray batch, cols[2];
    struct ArrowArray *children[2];
    const void *col_buffers[2][2];
    const void *batch_buffers[1];
} bid_batch;

static void make_batch(bid_batch *b, const int32_t *auction, const int32_t *price, int64_t rows, int64_t offset) {
    int c;
    memset(b, 0, sizeof *b);
    for (c = 0; c < 2; ++c) {
        b->col_buffers[c][0] = NULL; /* no validity bitmap: NEXMark fields are non-nullable */
        b->col_buffers[c][1] = c == 0 ? (const void *)auction : (const void *)price;
        b->cols[c].length = rows + offset;
        b->cols[c].n_buffers = 2;
        b->cols[c].buffers = b->col_buffers[c];
        b->cols[c].release = noop_release_array;
        b->children[c] = &b->cols[c];
    }
    b->batch_buffers[0] = NULL;
    b->batch.length = rows;
    b->batch.offset = offset; /* a sliced batch: rows [offset, offset + rows) of the children */
    b->batch.n_buffers = 1;
    b->batch.buffers = b->batch_buffers;
    b->batch.n_children = 2;
    b->batch.children = b->children;
    b->batch.release = noop_release_array;
}

static int fail(flockgpu_ctx *ctx, const char *what, int rc) {
    fprintf(stderr, "%s: status %d: %s\n", what, rc, ctx ? flockgpu_last_error(ctx) : "(no ctx)");
    return 1;
}

int main(int argc, char **argv) {
    flockgpu_ctx *ctx = NULL;
    flockgpu_plan *plan = NULL;
    struct ArrowSchema schema, fields[2], *field_ptrs[2], out_schema;
    struct ArrowArray out;
    bid_batch b0, b1;
    const struct ArrowArray *batches[2];
    int32_t auction[64], price[64];
    char *json;
    long len;
    FILE *f;
    int rc, i, invocation, modulus;

    if (argc != 3) {
        fprintf(stderr, "usage: consumer <q2 plan.json> <modulus>\n");
        return 2;
    }
    modulus = atoi(argv[2]);
    f = fopen(argv[1], "rb");
    if (!f) return fail(NULL, "open plan", -1);
    fseek(f, 0, SEEK_END);
    len = ftell(f);
    fseek(f, 0, SEEK_SET);
    json = (char *)malloc((size_t)len + 1);
    if (!json || fread(json, 1, (size_t)len, f) != (size_t)len) return fail(NULL, "read plan", -1);
    fclose(f);

    if (flockgpu_abi_version() != FLOCKGPU_ABI_VERSION) return fail(NULL, "abi version", flockgpu_abi_version());
    if ((rc = flockgpu_ctx_create(0, NULL, &ctx)) != FLOCKGPU_OK) return fail(ctx, "ctx_create", rc);
    if ((rc = flockgpu_plan_create(ctx, json, (size_t)len, &plan)) != FLOCKGPU_OK) return fail(ctx, "plan_create", rc);
    if (flockgpu_plan_query(plan) != 2 || flockgpu_plan_num_inputs(plan) != 1 || strcmp(flockgpu_plan_input_name(plan, 0), "bid") != 0)
        return fail(ctx, "plan is not q2 over bid", -1);
    if (flockgpu_plan_is_shuffling(plan) || flockgpu_plan_output_partitions(plan) != 1) return fail(ctx, "q2 does not shuffle", -1);

    /* schema: struct<auction: int32, price: int32> */
    memset(&schema, 0, sizeof schema);
    memset(fields, 0, sizeof fields);
    fields[0].format = "i";
    fields[0].name = "auction";
    fields[1].format = "i";
    fields[1].name = "price";
    for (i = 0; i < 2; ++i) {
        fields[i].release = noop_release_schema;
        field_ptrs[i] = &fields[i];
    }
    schema.format = "+s";
    schema.name = "";
    schema.n_children = 2;
    schema.children = field_ptrs;
    schema.release = noop_release_schema;
    if (flockgpu_plan_input_matches(plan, 0, &schema) != 1) return fail(ctx, "schema does not match the leaf", -1);

    for (invocation = 0; invocation < 2; ++invocation) {
        for (i = 0; i < 64; ++i) {
            auction[i] = 984 + 41 * i * (invocation + 1); /* a multiple of 123 when i * (invocation + 1) is a multiple of 3 */
            price[i] = 7 * i + invocation;
        }
        make_batch(&b0, auction, price, 40, 0);      /* rows 0 .. 39 */
        make_batch(&b1, auction, price, 20, 40);     /* rows 40 .. 59 as a slice (Arrow `offset`) */
        batches[0] = &b0.batch;
        batches[1] = &b1.batch;
        if ((rc = flockgpu_plan_feed(plan, 0, &schema, batches, 2)) != FLOCKGPU_OK) return fail(ctx, "plan_feed", rc);
        memset(&out, 0, sizeof out);
        memset(&out_schema, 0, sizeof out_schema);
        if ((rc = flockgpu_plan_execute(plan, &out_schema, &out)) != FLOCKGPU_OK) return fail(ctx, "plan_execute", rc);
        if (out_schema.n_children != 2 || strcmp(out_schema.children[0]->name, "auction") != 0 || strcmp(out_schema.children[1]->format, "i") != 0)
            return fail(ctx, "unexpected output schema", -1);
        if (out.n_children != 2 || out.children[0]->length != out.length) return fail(ctx, "unexpected output batch", -1);
        {
            const int32_t *oa = (const int32_t *)out.children[0]->buffers[1] + out.children[0]->offset;
            const int32_t *op = (const int32_t *)out.children[1]->buffers[1] + out.children[1]->offset;
            int64_t r, expect = 0;
            for (i = 0; i < 60; ++i) expect += auction[i] % modulus == 0;
            if (out.length != expect) return fail(ctx, "row count differs from the plain C loop", (int)out.length);
            for (r = 0; r < out.length; ++r) printf("%d %d\n", (int)oa[r], (int)op[r]);
        }
        out.release(&out);               /* the consumer owns the batch (pinned host memory) and frees it through Arrow's callback */
        out_schema.release(&out_schema);
        if (out.release != NULL || out_schema.release != NULL) return fail(ctx, "release callbacks must mark the structs released", -1);
        if ((rc = flockgpu_plan_reset(plan)) != FLOCKGPU_OK) return fail(ctx, "plan_reset", rc);
        printf("--\n");
    }
    /* an unfed plan is an empty relation (context.rs:305-314): zero rows, same schema */
    memset(&out, 0, sizeof out);
    memset(&out_schema, 0, sizeof out_schema);
    if ((rc = flockgpu_plan_execute(plan, &out_schema, &out)) != FLOCKGPU_OK || out.length != 0) return fail(ctx, "empty execute", rc);
    out.release(&out);
    out_schema.release(&out_schema);
    /* errors are statuses with a message, never aborts */
    if (flockgpu_plan_feed(plan, 3, &schema, batches, 2) != FLOCKGPU_ERR_INVALID || strlen(flockgpu_last_error(ctx)) == 0)
        return fail(ctx, "a bad input index must be FLOCKGPU_ERR_INVALID with a message", -1);
    /* round-4 surface, from C as a Rust host would drive it: the pane ring (one pane per invocation, the window = the panes held),
     * an asynchronous execute waited for later, and the guarded allocation */
    {
        int pane, n_parts = 0, held = 0, ppw = 0;
        int64_t first_pane = -1;
        void *guarded = NULL;
        if ((rc = flockgpu_plan_ring_open(plan, 2)) != FLOCKGPU_OK) return fail(ctx, "ring_open", rc);
        for (pane = 0; pane < 3; ++pane) {
            int64_t expect = 0;
            for (i = 0; i < 64; ++i) {
                auction[i] = 123 * (i + 1 + 64 * pane) * ((i % 4) == 0 ? 1 : 0) + ((i % 4) == 0 ? 0 : 1);   /* every fourth row passes */
                price[i] = 1000 * pane + i;
            }
            make_batch(&b0, auction, price, 64, 0);
            batches[0] = &b0.batch;
            if ((rc = flockgpu_plan_feed_pane(plan, 0, (int64_t)pane, &schema, batches, 1)) != FLOCKGPU_OK) return fail(ctx, "feed_pane", rc);
            if ((rc = flockgpu_plan_ring_state(plan, &first_pane, &held, &ppw)) != FLOCKGPU_OK) return fail(ctx, "ring_state", rc);
            if (ppw != 2 || held != (pane == 0 ? 1 : 2) || first_pane != (pane == 0 ? 0 : pane - 1)) return fail(ctx, "ring holds the wrong panes", held);
            memset(&out, 0, sizeof out);
            memset(&out_schema, 0, sizeof out_schema);
            if ((rc = flockgpu_plan_execute_async(plan, 0)) != FLOCKGPU_OK) return fail(ctx, "execute_async", rc);
            if ((rc = flockgpu_plan_wait(plan, &out_schema, &out, 1, &n_parts)) != FLOCKGPU_OK || n_parts != 1) return fail(ctx, "plan_wait", rc);
            expect = 16 * (pane == 0 ? 1 : 2);       /* the rows of the panes held that pass the filter (modulus 123) */
            if (modulus == 123 && out.length != expect) return fail(ctx, "ring window row count", (int)out.length);
            printf("ring pane %d rows %d\n", pane, (int)out.length);
            out.release(&out);
            out_schema.release(&out_schema);
            if ((rc = flockgpu_plan_reset(plan)) != FLOCKGPU_OK) return fail(ctx, "ring reset", rc);
        }
        /* pane 3 crosses the bus ahead of its turn (a host that reads ahead), then takes its place with no batches handed over */
        for (i = 0; i < 64; ++i) {
            auction[i] = 123 * (i + 1);              /* every row passes */
            price[i] = 3000 + i;
        }
        make_batch(&b0, auction, price, 64, 0);
        batches[0] = &b0.batch;
        if ((rc = flockgpu_plan_prefetch_pane(plan, 0, 3, &schema, batches, 1)) != FLOCKGPU_OK) return fail(ctx, "prefetch_pane", rc);
        if (flockgpu_plan_prefetch_pane(plan, 0, 3, &schema, batches, 1) == FLOCKGPU_OK) return fail(ctx, "one prefetch at a time", -1);
        if ((rc = flockgpu_plan_feed_pane(plan, 0, 3, NULL, NULL, 0)) != FLOCKGPU_OK) return fail(ctx, "feed of the prefetched pane", rc);
        memset(&out, 0, sizeof out);
        memset(&out_schema, 0, sizeof out_schema);
        if ((rc = flockgpu_plan_execute(plan, &out_schema, &out)) != FLOCKGPU_OK) return fail(ctx, "execute over a prefetched pane", rc);
        if (modulus == 123 && out.length != 16 + 64) return fail(ctx, "prefetched window row count", (int)out.length);
        printf("ring pane 3 rows %d (prefetched)\n", (int)out.length);
        out.release(&out);
        out_schema.release(&out_schema);
        if ((rc = flockgpu_plan_reset(plan)) != FLOCKGPU_OK) return fail(ctx, "ring reset", rc);
        if (flockgpu_plan_feed_pane(plan, 0, 7, &schema, batches, 1) != FLOCKGPU_ERR_INVALID) return fail(ctx, "a skipped pane must be refused", -1);
        if ((rc = flockgpu_plan_ring_close(plan)) != FLOCKGPU_OK) return fail(ctx, "ring_close", rc);
        if ((rc = flockgpu_malloc_guarded(ctx, 4096, &guarded)) != FLOCKGPU_OK || !guarded || ((size_t)guarded & 15)) return fail(ctx, "malloc_guarded", rc);
        if ((rc = flockgpu_memcpy(ctx, guarded, auction, sizeof auction, FLOCKGPU_H2D)) != FLOCKGPU_OK) return fail(ctx, "copy into guarded memory", rc);
        if ((rc = flockgpu_free_guarded(ctx, guarded)) != FLOCKGPU_OK) return fail(ctx, "free_guarded", rc);
        if (flockgpu_plan_check_partition_scheme(flockgpu_plan_partition_scheme()) != FLOCKGPU_OK) return fail(ctx, "the library's own scheme must check", -1);
        if (flockgpu_plan_check_partition_scheme("datafusion/ahash-0000") == FLOCKGPU_OK) return fail(ctx, "a foreign scheme must not check", -1);
    }
    printf("partition scheme %s\n", flockgpu_plan_partition_scheme());
    flockgpu_plan_destroy(plan);
    flockgpu_ctx_destroy(ctx);
    free(json);
    return 0;
}
