#!/bin/bash
# round 6, call C: lab experiment -- working copies built from mapped physical chunks (EMG3D_PLACE_VMM), 2 fresh processes
mkdir -p gpurun_out/r06/c
for i in 1 2 3; do
  EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so EMG3D_PLACE_VMM=1 EMG3D_LOG_SETUP=1 python bench.py --workload 256V --no-cpu --no-tol --batch 0 --steps 3 --no-roofline > gpurun_out/r06/c/vmm_$i.json 2> gpurun_out/r06/c/vmm_$i.err
  grep -h "\[place" gpurun_out/r06/c/vmm_$i.err
done
