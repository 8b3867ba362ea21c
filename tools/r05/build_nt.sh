#!/bin/bash
# lab builds of the library with a cache-policy mask (common.hpp: EMG3D_NT) -> emg3d_amd/libemg3d_hip_nt<mask>.so (git-ignored, travels with gpurun)
cd "$(dirname "$0")/../.."
n=0
for m in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -DEMG3D_LAB -DEMG3D_NT=$m \
      -o emg3d_amd/libemg3d_hip_nt$m.so emg3d_amd/csrc/emg3d_hip.hip > /tmp/build_nt$m.log 2>&1; echo "nt$m rc=$?" ) &
  n=$((n+1)); if [ $((n % 4)) = 0 ]; then wait; fi
done
wait
