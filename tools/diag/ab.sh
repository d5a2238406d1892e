#!/bin/bash
# diagnostic A/B on ONE box: for every library variant in build_variants/ (made here with `cp csrc/liblvi_hip.so build_variants/NAME.so`)
# put it in place and run the serialized 1x8 kernel table and/or the quick headline.   usage: ab.sh [table|head|both] [bench args…]
MODE=${1:-both}; shift
LIB=lidar-visual-inertial-slam_amd/csrc/liblvi_hip.so
cp $LIB /tmp/lvi_keep.so
for v in build_variants/*.so; do
  n=$(basename $v .so); cp $v $LIB
  echo "######## variant $n"
  if [ $MODE != head ]; then bash tools/diag/vb_exp.sh LVI_VARIANT $n | grep -E "steady|vb_|vox_|icp_|feat_sector|grid_" || exit 1; fi
  if [ $MODE != table ]; then bash tools/diag/quick_headline.sh "$@" || exit 1; fi
done
cp /tmp/lvi_keep.so $LIB
