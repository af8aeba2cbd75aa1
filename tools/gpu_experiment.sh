#!/bin/bash
O=$1; mkdir -p $O
timeout 300 python tools/dropin_latency.py > $O/dropin_latency.txt 2>&1; tail -2 $O/dropin_latency.txt
