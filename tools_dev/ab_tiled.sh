python tools_dev/bench_layer.py deconv 8 96 64 80 16 8 3 1 1 10
python tools_dev/bench_layer.py deconv 8 48 32 40 32 16 3 1 1 10
python tools_dev/bench_layer.py deconv 8 24 16 20 64 32 3 1 1 10
python tools_dev/bench_layer.py 3d 8 96 64 80 16 16 3 1 1 10
python tools_dev/bench_layer.py 3d 4 192 128 160 8 16 3 1 1 10
python tools_dev/bench_layer.py 3d 8 48 32 40 32 32 3 1 1 10
python tools_dev/bench_layer.py 3d 8 24 16 20 64 64 3 1 1 10
