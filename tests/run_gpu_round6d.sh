#!/bin/bash
# round 6: the f16mx stem's LDS pattern A/B (product against -DOIBL_STEM_R5_LDS debug library): timing, roles, PMC.
cd "$(dirname "$0")/.."
R=$(pwd)
TAG=${1:-r06_d}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python tests/gpu_stem_lds_ab.py 5 2>&1 | grep -v amdgpu.ids | tee $OUT/stem_lds_ab.txt
timeout 300 python tests/gpu_stem_mx_bench.py 0 2>&1 | grep -v amdgpu.ids | tee -a $OUT/stem_lds_ab.txt
timeout 600 python -m pytest tests/test_gpu_api.py -q -x -k "groups or default_precision or settles" -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/pytest_item7.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc1 -o s -- python $R/tests/gpu_stem_lds_ab.py 1 > $OUT/pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $OUT/pmc2 -o s -- python $R/tests/gpu_stem_lds_ab.py 1 > $OUT/pmc2.log 2>&1
tail -2 $OUT/pmc1.log $OUT/pmc2.log
cd $R
find $OUT -type f -size +8M -print -delete
find $OUT -name "*counter_collection.csv" | head
