cd $GRAFT_REPO_ROOT
python scripts/probes/two_stream_pair.py 2>&1 | tail -5
GPU_MAX_HW_QUEUES=4 python scripts/probes/two_stream_pair.py 2>&1 | tail -4
