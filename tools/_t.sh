cd $GRAFT_REPO_ROOT
for shp in "128,192,40,40,31" "128,128,48,48,47"; do for nb in 2 3; do echo "--- $shp NB=$nb"; TEAM_SHAPE=$shp SLAK_TEAM_NB=$nb SLAK_STREAM_TRI=0 timeout 300 python tools/time_team.py 2>&1 | grep -v amdgpu.ids | cut -c28- ; done; done
