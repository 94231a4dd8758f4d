python -m pytest tests/test_gpu_parity.py -x -q -k "parity_with_oracle_many_envs or native_library or joint" 2>&1 | tail -2
bash tools/gpu/bench16.sh
