import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'whisper.axera_amd/tools')
import whisper_axera_amd as wa, modelgen
mdir = '/tmp/axw_bench_models'
if not os.path.exists(mdir + '/small/small.safetensors'):
    modelgen.write_model_dir(mdir, 'small', seed=0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
e = wa.Whisper('small', mdir, 'zh', device=0, max_batch=B)
for what in ('decode_gemv',):
    ms = e.bench(what, B, 224, 50)
    print('dbg', os.environ.get('AXW_DEBUG_GEMM', '0'), what, 'us/launch %.2f' % (ms / 50 * 1e3 / 73))
