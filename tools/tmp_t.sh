python -m pytest tests/test_gpu_kv_narrow.py -x -q -k "do_not_fit" 2>&1 | grep -E "assert|Error|gens|shares" | head -20
