set -x
python -m pytest tests -m gpu -q -x -k "m512_full_size and (o or gate_up)" 2>&1 | tail -5
for shape in "8192 8192" "57344 8192" "8192 28672" "10240 8192"; do set -- $shape
 for m in 16375 4314; do
  python tools/time_ids.py --m $m --n $1 --k $2 --fmt nv auto 124c146113101008 124c146113101004 2>&1 | grep -v amdgpu.ids
  python tools/time_ids.py --m $m --n $1 --k $2 --fmt mx auto 124c146123101008 124c146123101004 2>&1 | grep -v amdgpu.ids
 done
done
