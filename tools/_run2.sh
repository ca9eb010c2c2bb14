cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r3b; mkdir -p $o
for c in C Chard; do
  timeout -k 10 300 python tools/render_probe.py $c --kernels 16,4,2,1 > $o/probe_$c.txt 2>&1 || { echo "probe $c failed"; tail -5 $o/probe_$c.txt; exit 1; }
  GS_LIB_OVERRIDE=$PWD/build_variants/lib_single.so timeout -k 10 300 python tools/render_probe.py $c --kernels 16 --no-stats > $o/probe_${c}_single.txt 2>&1 || { echo "probe single $c failed"; tail -5 $o/probe_${c}_single.txt; exit 1; }
done
timeout -k 10 200 python tools/render_probe.py D --kernels 0,16 --no-stats > $o/probe_D.txt 2>&1
cat $o/probe_C.txt $o/probe_C_single.txt $o/probe_Chard.txt $o/probe_Chard_single.txt $o/probe_D.txt | grep -v amdgpu.ids
