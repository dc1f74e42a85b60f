for lb in 2 3 4; do
  for c in "CSTR {\"n_robust\":0,\"collocation_deg\":3}" "oscillating_masses {}" "batch_reactor {\"n_horizon\":50}"; do
    set -- $c
    DOMPC_LB=$lb timeout 600 python tools/gpu_time_case.py $1 "$2" 16384 2>&1 | tail -1 | sed "s/^/LB=$lb /"
  done
done
