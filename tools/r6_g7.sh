cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for v in 1 2 4 8 16 34 63; do echo "variant $v"; SS_AMD_LIBRARY=$GRAFT_REPO_ROOT/tools/bin/ctcdbg/lib_$v.so timeout 200 python tools/ctc_probe.py 2>&1 | grep utterances; done
echo base; timeout 200 python tools/ctc_probe.py 2>&1 | grep utterances
