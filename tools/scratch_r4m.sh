mkdir -p gpurun_out/r4m
echo "256x256, 2 layers (216 tiles):"; SAVIT_GROUP_TILE=256 python tools/bench_wgrad_group.py 768 3072 25216 2 2>&1 | grep round | sed 's/per-weight launches.*| grouped/grouped/'
echo "640 mixed, 3 layers (216 big tiles):"; SAVIT_GROUP_TILE=640 python tools/bench_wgrad_group.py 768 3072 25216 3 2>&1 | grep round | sed 's/per-weight launches.*| grouped/grouped/'
echo "384 (128x384), 1 layer (144 small tiles):"; SAVIT_GROUP_TILE=384 python tools/bench_wgrad_group.py 768 3072 25216 1 2>&1 | grep round | sed 's/per-weight launches.*| grouped/grouped/'
echo "ViT-L 256x256, 1 layer (192 tiles):"; SAVIT_GROUP_TILE=256 python tools/bench_wgrad_group.py 1024 4096 147712 1 2>&1 | grep round | sed 's/per-weight launches.*| grouped/grouped/'
echo "ViT-L 640, 1 layer (132 big tiles):"; SAVIT_GROUP_TILE=640 python tools/bench_wgrad_group.py 1024 4096 147712 1 2>&1 | grep round | sed 's/per-weight launches.*| grouped/grouped/'
