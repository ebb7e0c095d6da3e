set -e
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline | tail -1 | cut -c1-400
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 3 --warmup 1 --mode slab --size 8192 --sweeps-per-step 240 | tail -1 | cut -c1-400
python bench.py --steps 3 --warmup 1 --batch 16 --size 1024 --no-cpu-baseline | tail -1 | cut -c1-300
