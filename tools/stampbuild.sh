#!/bin/bash
# Build a diagnostic copy of the library with in-kernel cycle stamps (tools/_stamp/, never shipped or committed).
# usage: tools/stampbuild.sh -DVSLAM_POSE_STAMPS [-D...]
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_stamp
for f in gtsam-vslam_amd/csrc/*.hip; do
  o=tools/_stamp/$(basename ${f%.hip}).o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt "$@" -c $f -o $o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_stamp/libvslam_stamp.so tools/_stamp/*.o -lpthread -ldl
