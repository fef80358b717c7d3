#!/bin/bash
O=gpurun_out/r03_24; mkdir -p $O
python scripts/probes/amp_table_diag.py > $O/diag.txt 2>&1; tail -20 $O/diag.txt
