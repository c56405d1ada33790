cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_scale.py -x -q -m gpu -k "config3_full_size_parity and fp32" 2>&1 | grep -v "warning: loop not unrolled\|^ *[0-9]* |\|^$" | grep -B30 -A12 "^E " | cut -c1-260 | tail -70
( time timeout 1500 python -m pytest tests/test_launcher.py tests/test_hip_step.py tests/test_hip_dropin.py -x -q -m gpu 2>&1 | tail -25 ) 2>&1 | cut -c1-200
