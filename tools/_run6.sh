for nw in 8 4; do for dg in 0 1; do
  echo "=== NW=$nw DIAG=$dg"
  BRIEF_K16_NW=$nw BRIEF_DIAG=$dg python3 tools/read_stamps16.py 9 512 2>&1 | grep -v amdgpu
done; done
