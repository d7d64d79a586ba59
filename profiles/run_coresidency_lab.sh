#!/bin/bash
# r06: the co-residency A/Bs on one box (libraries from profiles/coresidency_lab_build.py).  Output: gpurun_out/lab_coresidency.txt
mkdir -p gpurun_out
OUT=gpurun_out/lab_coresidency.txt
: > $OUT
L=profiles/_lab
run() { echo "== $*" >> $OUT; ( "$@" ) >> $OUT 2>&1; echo "   rc $?" >> $OUT; }
for lib in $L/libt2h_select.so $L/libt2h_select_pad4.so ""; do
    T2H_LIBRARY=$lib run timeout 300 python profiles/coresidency_repro.py ${ROUNDS:-50}
done
T2H_LIBRARY= run timeout 600 python profiles/coresidency_trunk_first.py ${LAUNCHES:-2000} bx3 3
T2H_LIBRARY=$L/libt2h_trunk_pad4.so run timeout 600 python profiles/coresidency_trunk_first.py ${LAUNCHES:-2000} bx3 3
T2H_LIBRARY= run timeout 600 python profiles/coresidency_trunk_first.py ${LAUNCHES:-2000} mfma 3
tail -60 $OUT
