timeout 200 python tools/time_configs.py 2>&1 | grep cfg
F3DS_LIB=$PWD/fast-3d-pointcloud-segmentation_amd/libf3ds_4b486e7.so timeout 200 python tools/time_configs.py 2>&1 | grep cfg
