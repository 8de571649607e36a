for v in "" wexp1 wexp2 wexp4 wexp8; do
  if [ -z "$v" ]; then unset ADAMVS_LIB_PATH; else export ADAMVS_LIB_PATH=ada-mvs_amd/libadamvs_hip.$v.so; fi
  python3 bench.py --no-cpu-baseline --no-cascade --steps 5 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg2 b128 variant [$v]', round(d['ms_per_step'],2), d['phase_ms_per_step']['s1.recurrence'])"
done
