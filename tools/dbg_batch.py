import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fewshot_vit_amd import synthetic, models
from fewshot_vit_amd.utils import few_shot as fs
from oracle import visformer_oracle as vo
cfg = vo.VisformerCfg()
shapes = vo.state_dict_shapes(cfg, prefix='encoder.'); shapes['temp'] = ()
sd = synthetic.synthetic_checkpoint_sd(shapes)
m = models.make('meta-baseline', encoder='visformer_micro_80', encoder_args={'numerics': 'bf16'})
m.load_state_dict(sd, strict=True); m = m.cuda().eval()
x = synthetic.synthetic_episodes(77, 3, 5, 1, 15)
xs, xq = fs.split_shot_query(x, 5, 1, 15, 3)
with torch.no_grad():
    a1 = m(xs.cuda(), xq.cuda()).cpu()
    a2 = m(xs.cuda(), xq.cuda()).cpu()
    s1 = torch.cat([m(xs[e:e + 1].cuda(), xq[e:e + 1].cuda()).cpu() for e in range(3)])
    s2 = torch.cat([m(xs[e:e + 1].cuda(), xq[e:e + 1].cuda()).cpu() for e in range(3)])
print('batched repeat equal', torch.equal(a1, a2), 'single repeat equal', torch.equal(s1, s2))
d = (a1 - s1).abs()
print('max diff', d.max().item(), 'per-episode', d.amax(dim=(1, 2)))
# features
enc = m.encoder
imgs = torch.cat([xs.flatten(0, 2), xq.flatten(0, 1)]).cuda()
with torch.no_grad():
    f_all = enc(imgs).cpu()
    f_80 = enc(imgs[:80]).cpu()
    f_40 = enc(imgs[:40]).cpu()
print('feat diff 80 vs all', (f_all[:80] - f_80).abs().max().item(), 'rows differing', ((f_all[:80] - f_80).abs().amax(1) > 0).nonzero().flatten().tolist()[:40])
print('feat diff 40 vs all', (f_all[:40] - f_40).abs().max().item(), ((f_all[:40] - f_40).abs().amax(1) > 0).nonzero().flatten().tolist()[:40])
