bash scripts/probe/v9_pmc_ablate.sh gpurun_out/r05/b2 nodma noread nodma_noread nobar
