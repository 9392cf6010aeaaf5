export PYTHONPATH=.
for sh in "0 0 2276 1024 4096" "0 0 2276 1024 6048" "0 0 1300 1024 4096" "0 0 3500 1024 4096"; do
for env in "X=1" "PK2_GEMM_SPLITK=1"; do
  echo "$env: $(env $env python tools/dbg/gemm_one.py $sh 2>&1 | grep -v amdgpu)"
done; done
