cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02g_wf -o wf -- python3 $GRAFT_REPO_ROOT/workflows/mapmaker_pcg.py > $GRAFT_REPO_ROOT/gpurun_out/r02g_wf.log 2>&1
cd $GRAFT_REPO_ROOT
python workflows/mapmaker_pcg.py > gpurun_out/r02g_wf_plain.log 2>&1
