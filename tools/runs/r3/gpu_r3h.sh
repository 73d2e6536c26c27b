# Round 3: sparse pair kernel workgroup-size variants; whole gpu suite with the defaults.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
export TMPDIR=/tmp
b() { name=$1; shift; "$@" > gpurun_out/r3h/$name.json 2> gpurun_out/r3h/$name.err; python -c "
import json; d=json.load(open('gpurun_out/r3h/$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],4)); print('   ', {k:round(v['ms']*1000,1) for k,v in d['kernels'].items()})"; }
for t in base P512 P1024 P512N2 P256N2T4; do
lib=$PWD/visual_marker_mapping_amd/libvmm_ba_$t.so; [ $t = base ] && lib=$PWD/visual_marker_mapping_amd/libvmm_ba.so
VMM_BA_LIB=$lib VMM_BA_SCHUR=sparse b sparse_v0.25_$t timeout -k 10 300 python bench.py --no-cpu-baseline --visibility 0.25 --steps 70
VMM_BA_LIB=$lib VMM_BA_SCHUR=sparse b sparse_v0.5_$t timeout -k 10 300 python bench.py --no-cpu-baseline --visibility 0.5 --steps 70
done
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3h/pytest_gpu.txt 2>&1; tail -5 gpurun_out/r3h/pytest_gpu.txt
