cd $GRAFT_REPO_ROOT
export CHX_ALLOW_STALE_LIB=1  # variant builds carry no source hash
run() { # wl steps
  python scripts/rebuild_profile.py $1 $2 2>&1 | grep -v amdgpu.ids | grep "$1:\|inner" | tr '\n' ' '; echo; }
for cfg in "0.08 16" "0.10 24" "0.12 32" "0.14 40" "0.16 48" "0.20 64"; do
  set -- $cfg
  echo "== inner_skin=$1 every=$2"
  CHX_INNER_SKIN=$1 CHX_PRUNE_EVERY=$2 run c3 3000
  CHX_INNER_SKIN=$1 CHX_PRUNE_EVERY=$2 run c4 1500
done
echo "== G variants at (0.12, 32)"
for g in 4 8 16; do echo -n "c3 G=$g: "; CHX_FL_G=$g run c3 3000; done
for g in 2 4 8; do echo -n "c4 G=$g: "; CHX_FL_G=$g run c4 1500; done
