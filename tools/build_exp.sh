# experiments library (agd_bench_* entry points, AGD_IGEMM_LOG, timing knobs): agenda_amd/libagenda_hip_exp.so.  Build it in the
# container before a gpurun call (in-tree .so files travel to the GPU box) or on the box itself.
make -C "$(dirname "$0")/../agenda_amd/csrc" -j8 exp
