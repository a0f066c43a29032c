#!/bin/bash
# round 5: workgroup durations and the shader clock of the wide vocabulary kernel's ablations (-DJLM_WGTIME builds; zero operands = no power effects)
mkdir -p gpurun_out
O=gpurun_out/r05_e_clock2.txt; : > $O
for lib in T33 T97 T37 T45 T33M2; do
    echo "zero=1 wide lib=$lib (33 no fold + no fragment reads; 97 = 33 + one operand set; 37 = 33 + no DMA; 45 = 37 + no barrier; T33M2: k = 100 with tiles of 2 blocks):" >> $O
    KBENCH_ZERO=1 JLM_MX_WIDE=1 JLM_HIP_LIB=$PWD/build_prof/libjlm_hip_$lib.so timeout 200 python tools/probes/mixed_clock.py 200 100 2>&1 | grep "k=" >> $O
done
cat $O
