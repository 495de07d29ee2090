cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in a b; do cp tools/_lib_$v.so slak_amd/lib/libslak_hip.so; echo "--- variant $v (a: io prio 3, b: none)"; TEAM_SHAPES=2 SLAK_STREAM_TRI=0 timeout 300 python tools/time_team.py 2>&1 | grep -v amdgpu.ids | grep dgrad | cut -c40-; done; done
