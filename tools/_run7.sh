cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r3g; mkdir -p $o
timeout -k 10 1000 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "shader_main or readme_shape or config_d_4k or config_c_hard or library_before" > $o/tests.log 2>&1; rc=$?; tail -12 $o/tests.log; echo "tests rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 900 python tools/readme_shapes.py --frames 200 > $o/readme_shapes.json 2> $o/readme_shapes.err; echo "shapes rc=$?"; grep -v amdgpu $o/readme_shapes.err | tail -14
