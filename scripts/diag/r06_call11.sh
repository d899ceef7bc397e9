for t in 8 12 16 5; do
  echo "== copy threads $t"
  MOMLEVEL_AMD_COPY_THREADS=$t python scripts/example_call.py --reps 7 --source numpy 2>/dev/null | cut -c270-420
done
echo "== download ring 6"
MOMLEVEL_AMD_DOWNLOAD_RING=6 python scripts/example_call.py --reps 7 --source numpy 2>/dev/null | cut -c270-420
echo "== piece 32 MiB"
MOMLEVEL_AMD_STAGING_PIECE_MIB=32 python scripts/example_call.py --reps 7 --source numpy 2>/dev/null | cut -c270-420
