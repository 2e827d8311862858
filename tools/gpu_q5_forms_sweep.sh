PARITY="wavep1 wavep4" FORMS="wg wave1 wavep1 wavep4" TAG=q5_forms2 bash tools/gpu_q5_count_forms.sh
for n in 8 12 16; do echo "waves/cu $n"; FLOCKGPU_Q5_WAVES_PER_CU=$n ROUNDS="" FORMS="wavep1 wavep4" TAG=q5_forms2_$n bash tools/gpu_q5_count_forms.sh; done
