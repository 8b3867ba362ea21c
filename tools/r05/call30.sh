#!/bin/bash
# round 5, call 30: large arrays of a handle carved out of ONE block (lab knob EMG3D_SLAB_GB) against one hipMalloc each: launch time
# of the 256^3 level-0 sweeps in fresh processes
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
export EMG3D_HIP_LIB=$PWD/emg3d_amd/libemg3d_hip_lab.so
{
for p in 1 2 3; do
  echo "process $p, one hipMalloc per array"; timeout 900 python3 tools/r05/bimodal.py 256V 0 0 2>/dev/null
  echo "process $p, slab 40 GB"; EMG3D_SLAB_GB=40 timeout 900 python3 tools/r05/bimodal.py 256V 0 0 0 2>/dev/null
done
echo "slab 40 GB, pieces aligned to 1 GiB"; EMG3D_SLAB_GB=60 EMG3D_SLAB_ALIGN=1073741824 timeout 900 python3 tools/r05/bimodal.py 256V 0 0 2>/dev/null
} | tee $O/c30_slab.txt
