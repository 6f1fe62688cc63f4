import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'whisper.axera_amd/tools')
import whisper_axera_amd as wa, modelgen
mdir = '/tmp/axw_bench_models'
if not os.path.exists(mdir + '/small/small.safetensors'):
    modelgen.write_model_dir(mdir, 'small', seed=0)
e = wa.Whisper('small', mdir, 'zh', device=0, max_batch=1)
ms = e.bench('decode_gemv', 1, 224, 50)
print('dbg', os.environ.get('AXW_DEBUG_GEMV', '0'), 'gemv us/launch %.2f' % (ms / 50 * 1e3 / 73))
