#!/bin/bash
# same-box A/B of the merge loop: round 3's library (fast-3d-pointcloud-segmentation_amd/libf3ds_r3.so, built from commit 874fa7d into .ab/) against the current one
R=$PWD/fast-3d-pointcloud-segmentation_amd
for rep in 1 2; do
  for lib in libf3ds_r3.so libf3ds.so; do
    echo "== $lib lone frame (8 waves)"; F3DS_LIB=$R/$lib python3 tools/lone_frame.py 4 2>&1 | cut -c1-190 | tail -2
    echo "== $lib lone frame, 4 waves"; F3DS_MERGE_NW=4 F3DS_LIB=$R/$lib python3 tools/lone_frame.py 4 2>&1 | cut -c1-190 | tail -2
    echo "== $lib config 4"; F3DS_LIB=$R/$lib python3 tools/config4_frame.py 3 2>&1 | cut -c1-230 | tail -2
  done
done
