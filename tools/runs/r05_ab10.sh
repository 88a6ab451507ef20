set -u
O=gpurun_out/r05; mkdir -p $O
AB_LIB=tools/_ab/libpgtwin_tuning.so AB_CONFIGS=fst AB_ROUNDS=10 AB_VARIANTS="16 waves x 2 loads:PGT_TUNE_FST=8:2:1024;12 waves x 4 loads:PGT_TUNE_FST=8:4:768;24 waves x 2 loads:PGT_TUNE_FST=4:2:1536;32 waves x 2 loads:PGT_TUNE_FST=4:2:2048;12 waves x 2 loads:PGT_TUNE_FST=8:2:768" python tools/build_ab.py 1e8 1.25e8 1e9 > $O/build_ab_fst_geometry_with_rotation_2.md 2>&1; echo "rc=$?"; tail -n 19 $O/build_ab_fst_geometry_with_rotation_2.md
