#!/bin/bash
# (GPU box) merge stage of a lone 1M-point frame (8 waves, and 4 waves with everything in global memory) and of the 20M scene: the current library against libf3ds_prev.so, alternating
R=$PWD/fast-3d-pointcloud-segmentation_amd
for rep in 1 2 3; do for lib in libf3ds.so libf3ds_prev.so; do
  echo -n "$lib 8w: "; F3DS_LIB=$R/$lib python3 tools/lone_frame.py 4 2>&1 | tail -2 | awk '{print $(NF-1)}' | tr '\n' ' '
  echo -n " 4w-global: "; F3DS_MERGE_NW=4 F3DS_MERGE_KEYS=global F3DS_LIB=$R/$lib python3 tools/lone_frame.py 4 2>&1 | tail -2 | awk '{print $(NF-1)}' | tr '\n' ' '
  echo -n " config4 merge ms: "; F3DS_LIB=$R/$lib python3 tools/config4_frame.py 3 2>&1 | grep "^scene" | tail -2 | sed 's/.*labels): //' | awk '{print $6}' | tr '\n' ' '; echo
done; done
