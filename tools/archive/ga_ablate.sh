# ablations of the group-ahead 256 x 256 kernel (tools/ablate_wide.sh 4 8 12 1 2 32 47): bf16 x NVFP4, `o` at M = 16375
for abl in shipped 4 8 12 1 2 32 47; do
  if [ $abl = shipped ]; then unset PETIT_AMD_LIB; else export PETIT_AMD_LIB=$PWD/tools/ablate/wide/libpetit_abl_$abl.so; fi
  python tools/time_ids.py --m 16375 --n 8192 --k 8192 --fmt nv --tag abl$abl 124c146113101008 124c146113101004 2>&1 | grep -v amdgpu.ids
done
