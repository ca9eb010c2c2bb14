cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r3f; mkdir -p $o
for g in 0 1920 2048 2304 3072 4096; do
  echo "grid $g band D 51:68"; GS_SORT_GRID=$g timeout -k 10 200 python tools/sort_probe.py --config D --rows 51:68 --frames 100 2>&1 | grep -v amdgpu
done
for g in 0 16384 16640 20480; do
  echo "grid $g D full"; GS_SORT_GRID=$g timeout -k 10 200 python tools/sort_probe.py --config D --frames 60 2>&1 | grep -v amdgpu
done
for g in 0 6464 6912; do
  echo "grid $g C full"; GS_SORT_GRID=$g timeout -k 10 200 python tools/sort_probe.py --config C --frames 100 2>&1 | grep -v amdgpu
done
for g in 0 1728 1792 2048; do
  echo "grid $g B full"; GS_SORT_GRID=$g timeout -k 10 200 python tools/sort_probe.py --config B --frames 100 2>&1 | grep -v amdgpu
done
