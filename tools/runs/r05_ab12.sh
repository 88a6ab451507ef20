set -u
O=gpurun_out/r05; mkdir -p $O
AB_LIB=tools/_ab/libpgtwin_tuning.so AB_CONFIGS=ext AB_ROUNDS=10 AB_VARIANTS="8w x 4 loads:PGT_EXT_VARIANT_NOW=1;8w x 8 loads:PGT_EXT_VARIANT_NOW=2;8w x 16 loads:PGT_EXT_VARIANT_NOW=3;16w x 8 loads:PGT_EXT_VARIANT_NOW=6;16w x 4 loads:PGT_EXT_VARIANT_NOW=8;32w x 4 loads:PGT_EXT_VARIANT_NOW=10;4w x 16 loads:PGT_EXT_VARIANT_NOW=5" python tools/build_ab.py 5e7 1e8 1.25e8 2.5e8 1e9 > $O/build_ab_ext_geometry_with_rotation.md 2>&1; echo "rc=$?"; tail -n 41 $O/build_ab_ext_geometry_with_rotation.md
