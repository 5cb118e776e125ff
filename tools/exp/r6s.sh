#!/bin/bash
# the packed-fp32 high-half-select probe, for profiles/: forms, co-runners, alone
{
echo "# tools/exp/lds_reader_probe.py: v_pk_*_f32 forms, 16 operations per iteration x 2048 iterations x 512 threads x 16 workgroups x 60 launches, every result against the known value"
echo "## beside the pooled bf16-split forward (split_nt_kernel<96, 4, ..., 128, true, 1>, one workgroup per CU)"
python3 tools/exp/lds_reader_probe.py 53 49 48 50 51 52 54 55 57 58 56 60 61 63 64 65 66 62 2>&1 | grep -v amdgpu.ids | cut -c1-260
echo "## alone"
python3 tools/exp/lds_reader_probe.py --alone 53 50 54 58 2>&1 | grep -v amdgpu.ids | cut -c1-260
echo "## v_pk_add_f32 ... op_sel:[0,1] beside bare spinners (256 threads, no LDS)"
python3 tools/exp/lds_reader_probe.py --co=bf16s,bf16,f32,valu 54 2>&1 | grep -v amdgpu.ids | cut -c1-260
} > gpurun_out/pk_hi_select_probe.txt 2>&1
cat gpurun_out/pk_hi_select_probe.txt
