cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/ubench/light_clock.hip -o /tmp/light_clock && /tmp/light_clock
