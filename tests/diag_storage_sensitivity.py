"""Diagnostic (not a test; CPU only, oracle only): which part of the network makes bf16 STORAGE between the kernels cost 15 % relative L2 on
conv_blocks.0.conv1.weight's gradient?  Runs the full-size oracle step (oracle/model_ref + loss_ref, the same batch and weights as
tests/test_fullsize.py) four times -- f32; bf16 storage everywhere (the yardstick of the bf16 kernels); bf16 storage except ResBlock 0; except all
three ResBlocks -- and prints mel-L1 and the gradient figures of the conv-stack tensors against the f32 run.  Answers the round-3 verdict's question
"what would f32 storage for ResBlock 0 do to the bf16 mode" without building that mode.
usage: python tests/diag_storage_sensitivity.py [rows]     (rows: packed rows of the batch to keep, default all ~110; 24 runs in a few minutes on 8 cores)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import loss_ref, model_ref  # noqa: E402
from silent_speech_amd.synthetic import reference_size_batch  # noqa: E402
from tests.util import rel_l2_cos  # noqa: E402


def main():
    from silent_speech_amd.architecture import Model
    torch.manual_seed(0)
    m = Model(112, 80, 48, model_size=768, num_layers=6, dropout=0.0, compute_dtype=torch.float32)
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    batch = reference_size_batch(seed=0)
    if len(sys.argv) > 1:                                   # a prefix of the utterances worth ~rows packed rows
        rows, keep, acc = int(sys.argv[1]), 0, 0
        for t in batch['raw_emg']:
            acc += t.shape[0]; keep += 1
            if acc >= rows * 1600:
                break
        batch = {k: (v[:keep] if isinstance(v, (list, tuple)) else v) for k, v in batch.items()}
    torch.set_num_threads(os.cpu_count() or 1)
    orig = model_ref.resblock

    def run(storage, exempt=()):
        sd = {k: v.clone() for k, v in sd0.items()}
        for v in sd.values():
            if v.dtype == torch.float32:
                v.requires_grad_(True)

        def resblock(x, sdd, p, stride, training, running_out=None):
            if p in exempt:
                save = model_ref._STORAGE[0]; model_ref._STORAGE[0] = None
                try:
                    return orig(x, sdd, p, stride, training, running_out)
                finally:
                    model_ref._STORAGE[0] = save
            return orig(x, sdd, p, stride, training, running_out)
        model_ref.resblock = resblock
        try:
            xr = loss_ref.combine_fixed_length(batch['raw_emg'], 1600)

            def go():
                pr, ar = model_ref.model_forward(sd, xr, training=True, shift_r=3, running_out={})
                l, _ = loss_ref.dtw_loss_ref(pr, ar, batch)
                l.backward()
                return pr.detach(), float(l)
            if storage:
                with model_ref.bf16_storage():
                    pr, l = go()
            else:
                pr, l = go()
        finally:
            model_ref.resblock = orig
        return pr, l, {k: v.grad for k, v in sd.items() if v.dtype == torch.float32 and v.grad is not None}

    ref_pred, ref_loss, ref_g = run(False)
    watch = ['conv_blocks.0.conv1.weight', 'conv_blocks.0.conv2.weight', 'conv_blocks.1.conv1.weight', 'conv_blocks.2.conv2.weight', 'w_raw_in.weight',
             'transformer.layers.0.linear1.weight', 'transformer.layers.5.linear2.weight', 'w_out.weight']
    out = {}
    for name, ex in (('bf16 storage everywhere', ()), ('f32 inside ResBlock 0', ('conv_blocks.0',)),
                     ('f32 inside all ResBlocks', ('conv_blocks.0', 'conv_blocks.1', 'conv_blocks.2'))):
        pr, l, g = run(True, ex)
        row = {'mel_l1': float((pr - ref_pred).abs().mean()), 'loss_rel': abs(l - ref_loss) / abs(ref_loss)}
        for n in watch:
            row[n] = round(rel_l2_cos(g[n], ref_g[n])[0], 5)
        row['worst'] = max((rel_l2_cos(g[n], ref_g[n])[0], n) for n in g if 'relative_positional' not in n and not (n.endswith('.bias') and ('conv' in n or 'residual' in n)))
        out[name] = row
        print(name, json.dumps(row))
    return out


if __name__ == '__main__':
    main()
