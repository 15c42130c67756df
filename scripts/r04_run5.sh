set -u
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 1200 scripts/ab_bench.sh run5 "--steps 60 --warmup 5 --no-legs --no-cpu-baseline" \
  base:X=1 pad2:X=1@pad2 nobar:X=1@nobar pad2nobar:X=1@pad2nobar base_b:X=1 pad2_b:X=1@pad2 nobar_b:X=1@nobar pad2nobar_b:X=1@pad2nobar \
  occ6:EVAH_LDS_EXTRA=4096 occ5:EVAH_LDS_EXTRA=10000 occ4:EVAH_LDS_EXTRA=18000
