cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b_f; mkdir -p $O
hipcc -O2 --offload-arch=gfx950 -o /tmp/peak tools/micro/mfma_f64_peak.hip && timeout -k 10 60 /tmp/peak > $O/peak.txt 2>&1; cat $O/peak.txt
