export PYTHONPATH=.
for sh in "0 1 2300 512 2048" "0 1 96 512 2048" "0 0 2300 512 2048" "0 1 1150 512 2048"; do
for env in "X=1" "PK2_GEMM_SPLITK=1" "PK2_GEMM_SPLITK=1 PK2_GEMM_TILES=2" "PK2_GEMM_SPLITK=1 PK2_GEMM_TILES=1"; do
  echo "$env: $(env $env python tools/dbg/gemm_one.py $sh 2>&1 | grep -v amdgpu)"
done; done
