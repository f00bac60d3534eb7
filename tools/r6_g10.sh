cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
echo base; python tools/mel_probe.py 2>&1 | grep frames
for v in tools/bin/melab/*.so; do echo $v; SS_AMD_LIBRARY=$GRAFT_REPO_ROOT/$v python tools/mel_probe.py 2>&1 | grep frames; done
