#!/bin/bash
# MFMA-shape power probe (scripts/probe/mfma_power.hip): throughput = sustained clock
scripts/probe/_probe_mfma_power 256 20000 8
MP_ZEROS=1 scripts/probe/_probe_mfma_power 256 20000 8
