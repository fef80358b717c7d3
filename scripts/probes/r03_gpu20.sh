#!/bin/bash
O=gpurun_out/r03_20; mkdir -p $O
python scripts/probes/post_timeline.py > $O/post_timeline.txt 2>&1; cat $O/post_timeline.txt | tail -40
