#!/bin/bash
# Per-kernel register / LDS / scratch usage of one .hip file (device-only assembly, gfx950): tools/kernel_resources.sh flock_amd/csrc/q8.hip [filter]
f=$1; pat=${2:-.}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-gpu-rdc --cuda-device-only -S -o /tmp/kres.s "$f" 2>/dev/null || exit 1
python3 - "$pat" <<'PY'
import re, sys
txt = open("/tmp/kres.s").read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    if not re.search(sys.argv[1], name): continue
    g = lambda k: (re.search(r"\.amdhsa_" + k + r" (\S+)", body) or [None, "?"])[1]
    short = re.sub(r"^_ZN\d+_GLOBAL__N_1", "", name)[:70]
    print(f"{short:70s} vgpr {g('next_free_vgpr'):>4s} sgpr {g('next_free_sgpr'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>5s}")
PY
