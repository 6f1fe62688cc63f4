import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'whisper.axera_amd/tools')
import whisper_axera_amd as wa, modelgen
mdir = '/tmp/axw_bench_models'
if not os.path.exists(mdir + '/small/small.safetensors'):
    modelgen.write_model_dir(mdir, 'small', seed=0)
for B in (1, 16, 64):
    e = wa.Whisper('small', mdir, 'zh', device=0, max_batch=B)
    ms = e.bench('encoder', B, 0, 5) / 5
    print('B', B, 'encoder ms %.3f' % ms, 'TFLOP/s %.1f' % (386.63e9 * B / (ms * 1e-3) / 1e12))
    e.close()
