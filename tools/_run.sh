D=$PWD/fast-3d-pointcloud-segmentation_amd
python tools/ab_merge.py $D/libf3ds.so $D/libf3ds_base.so 4 | tee gpurun_out/r2ad_ab.log
