set -e
python tools/kbench.py --sizes 128 --sweeps 480 --rounds 3 --variants "matfree_tb:nimg=64,matfree_tb:nimg=512,matfree_tb:nimg=1024,matfree_tb:nimg=2048,matfree_tb:nimg=3072,matfree_tb:nimg=3072:tb_LY=128,matfree_tb:nimg=4096,matfree_tb:nimg=4096:tb_LY=128"
python tools/kbench.py --sizes 256 --sweeps 480 --rounds 3 --variants "matfree_tb:nimg=256,matfree_tb:nimg=256:tb_LY=256,matfree_tb:nimg=1024,matfree_tb:nimg=1024:tb_LY=256,matfree_tb:nimg=1024:tb_LY=128"
python tools/kbench.py --sizes 512 --sweeps 480 --rounds 3 --variants "matfree_tb:nimg=64,matfree_tb:nimg=64:tb_LY=128,matfree_tb:nimg=256,matfree_tb:nimg=256:tb_LY=256"
python tools/kbench.py --sizes 1024 --sweeps 480 --rounds 3 --variants "matfree_tb,matfree_tb:nimg=16,matfree_tb:nimg=16:tb_LY=128,matfree_tb:nimg=64,matfree_tb:nimg=64:tb_LY=256"
python tools/kbench.py --sizes 4096 --sweeps 480 --rounds 3 --variants "matfree_tb,matfree_tb:nimg=4,matfree_tb:nimg=4:tb_LY=128"
