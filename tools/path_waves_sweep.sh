cd $GRAFT_REPO_ROOT
for w in 5120 3072 2048 1024; do echo "== MOONRT_PATH_WAVES=$w"; MOONRT_PATH_WAVES=$w python tools/rank_balance.py 8 2>&1 | grep world; done
for g in 1 2; do echo "== MOONRT_PATH_GRP=$g"; MOONRT_PATH_GRP=$g python tools/rank_balance.py 8 2>&1 | grep world; done
echo "== nsub 2"; MOONRT_PATH_NSUB=2 python tools/rank_balance.py 8 2>&1 | grep world
