bash tools/final_profiles.sh r05 pmc > gpurun_out/final_pmc.log 2>&1; tail -5 gpurun_out/final_pmc.log; ls gpurun_out/final | head -20
