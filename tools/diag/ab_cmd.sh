#!/bin/bash
# diagnostic A/B on ONE box: for every library variant in build_variants/ put it in place and run the given command
LIB=lidar-visual-inertial-slam_amd/csrc/liblvi_hip.so
cp $LIB /tmp/lvi_keep.so
for rep in 1 2; do
for v in build_variants/*.so; do
  cp $v $LIB; echo "######## variant $(basename $v .so) (pass $rep)"; "$@" || exit 1
done
done
cp /tmp/lvi_keep.so $LIB
