#!/bin/bash
# randomised parity campaigns on the final code of round 6 (product library: the launch selection as shipped, chain form on 4-block lines;
# then the lab library: the campaigns' kernel overrides take effect)
mkdir -p gpurun_out/r06/fuzz
for lib in prod lab; do
  if [ $lib = lab ]; then export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so; fi
  timeout 1500 python tests/tools/fuzz_parity.py ${FUZZ_N:-120} ${FUZZ_SEED:-606} > gpurun_out/r06/fuzz/parity_$lib.txt 2>&1; echo "fuzz_parity $lib rc $?"; tail -4 gpurun_out/r06/fuzz/parity_$lib.txt
  timeout 600 python tests/tools/fuzz_kernels.py ${FUZZ_NK:-150} ${FUZZ_SEED:-606} > gpurun_out/r06/fuzz/kernels_$lib.txt 2>&1; echo "fuzz_kernels $lib rc $?"; tail -3 gpurun_out/r06/fuzz/kernels_$lib.txt
  timeout 600 python tests/tools/fuzz_reuse.py ${FUZZ_NR:-30} ${FUZZ_SEED:-606} > gpurun_out/r06/fuzz/reuse_$lib.txt 2>&1; echo "fuzz_reuse $lib rc $?"; tail -3 gpurun_out/r06/fuzz/reuse_$lib.txt
done
