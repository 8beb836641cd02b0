#!/bin/bash
bash scripts/pmc_configs.sh q3,q3v "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" k1
bash scripts/pmc_configs.sh q3,q3v "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" k2
bash scripts/pmc_configs.sh q3,q3v "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM" k3
