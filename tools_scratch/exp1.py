import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'whisper.axera_amd/tools')
import whisper_axera_amd as wa, modelgen
mdir = '/tmp/axw_bench_models'
if not os.path.exists(mdir + '/small/small.safetensors'):
    modelgen.write_model_dir(mdir, 'small', seed=0)
e = wa.Whisper('small', mdir, 'zh', device=0, max_batch=1)
for what in ('decode_step', 'decode_gemv', 'decode_attn'):
    for arg in (10, 224, 440):
        ms = e.bench(what, 1, arg, 50)
        print(os.environ.get('AXW_DEBUG_SAME_LAYER_WEIGHTS', '-'), what, arg, 'us/step %.1f' % (ms / 50 * 1e3))
