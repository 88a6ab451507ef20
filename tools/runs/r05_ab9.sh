set -u
O=gpurun_out/r05; mkdir -p $O
AB_LIB=tools/_ab/libpgtwin_tuning.so AB_CONFIGS=fst AB_ROUNDS=10 AB_VARIANTS="8 loads:PGT_TUNE_FST=16:8:512;16 loads:PGT_TUNE_FST=16:16:512;16 waves x 4 loads:PGT_TUNE_FST=8:4:1024;16 waves x 8 loads:PGT_TUNE_FST=8:8:1024;2 loads:PGT_TUNE_FST=16:2:512;two waves per tile:PGT_TUNE_FST_PAIR=4" python tools/build_ab.py 5e7 1e8 1.25e8 2.5e8 1e9 > $O/build_ab_fst_geometry_with_rotation.md 2>&1; echo "rc=$?"; tail -n 36 $O/build_ab_fst_geometry_with_rotation.md
