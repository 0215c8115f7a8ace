cd $GRAFT_REPO_ROOT
export CHX_ALLOW_STALE_LIB=1  # variant builds carry no source hash
for wl in c2 c3 c4 c3_liquid c4_liquid; do
  steps=2000; [ $wl = c4_liquid ] && steps=300; [ $wl = c3_liquid ] && steps=1000
  for k in 0 32; do echo -n "K=$k "; CHX_PRUNE_EVERY=$k python scripts/rebuild_profile.py $wl $steps 2>&1 | grep -v amdgpu.ids | grep "$wl:\|inner" | tr '\n' ' '; echo; done
done
