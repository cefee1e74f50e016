D=$PWD/fast-3d-pointcloud-segmentation_amd
python tools/ab_merge.py $D/libf3ds_base.so $D/libf3ds_v2.so $D/libf3ds_v3.so $D/libf3ds_v4.so 3 | tee gpurun_out/r2ae_ab.log
