import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'whisper.axera_amd/tools')
import whisper_axera_amd as wa, modelgen
mdir = '/tmp/axw_bench_models'
if not os.path.exists(mdir + '/small/small.safetensors'):
    modelgen.write_model_dir(mdir, 'small', seed=0)
e = wa.Whisper('small', mdir, 'zh', device=0, max_batch=1)
for arg in (30, 224, 440):
    print('cross', os.environ.get('AXW_SPLIT_CROSS', '-'), 'self', os.environ.get('AXW_SPLIT_SELF', '-'), 'step', arg,
          'us/step %.1f' % (e.bench('decode_step', 1, arg, 50) / 50 * 1e3), 'attn %.1f' % (e.bench('decode_attn', 1, arg, 50) / 50 * 1e3))
