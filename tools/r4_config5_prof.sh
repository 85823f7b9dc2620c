# round 4: kernel statistics of BASELINE config 5 on one GPU (256^3 Kuhn tets, GMRES + field-split): where its 216 ms go
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c5prof; rm -rf $O; mkdir -p $O
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python3 tools/config5_probe.py 256 > $O/out.txt 2> $O/ks.err || { tail -5 $O/ks.err; exit 1; }
cp $(find $O/ks -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv; rm -rf $O/ks
cat $O/out.txt | grep -v amdgpu
head -16 $O/kernel_stats.csv | cut -d, -f1-5 | cut -c1-60,150-
