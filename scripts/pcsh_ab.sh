# usage (GPU box): bash scripts/pcsh_ab.sh -- the 8_16 scan passes with 4 levels x 16 segments x 32 | 16 columns per workgroup (round 4's
# default: one workgroup of 512 threads per CU at 228 VGPRs) against 2 levels x 32 segments x 8 | 16 columns (round 6: 168 VGPRs, three
# workgroups of 256 threads per CU), config 5's domain, same box
cd $GRAFT_REPO_ROOT
SOLVER=8_16 CFGS="4,16,32;4,16,16;2,32,8;2,32,16;4,16,32" python3 scripts/pcsbench.py 2>&1 | grep "^{"
