cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in base fft5_4 fft5_5 fft6_6 fft8_8; do
  L=$PWD/build/var/libbsrnn_$v.so; [ $v = base ] && L=$PWD/speechseparation_amd/lib/libbsrnn_hip.so
  BSRNN_HIP_LIB=$L python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-exact-f32 --no-train-step --no-in-flight 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['stages']
print('$v', d['ms_per_step'], 'stft=%.4f istft=%.4f' % (s['stft']['ms_per_step'], s['istft']['ms_per_step']))"
done; done
