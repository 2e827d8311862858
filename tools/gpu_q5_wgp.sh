#!/bin/bash
# the persistent workgroup form of the count pass (raw LDS barriers, next tile's loads in flight): FLOCKGPU_Q5_COUNT=wgp at 4 / 5 / 6 / 8 workgroups per CU
PARITY="wgp" FORMS="wg wgp" TAG=q5_wgp bash tools/gpu_q5_count_forms.sh
for n in 4 6 8; do echo "workgroups/cu $n"; FLOCKGPU_Q5_WAVES_PER_CU=$n ROUNDS="" FORMS="wgp" TAG=q5_wgp_$n bash tools/gpu_q5_count_forms.sh; done
