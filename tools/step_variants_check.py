import sys, torch
sys.path.insert(0, '.')
import vit_gan_amd
from vit_gan_amd.config import Config
from vit_gan_amd.engine import GanEngine
from vit_gan_amd.generator import SirenGenerator
from vit_gan_amd.modules import ViTDiscriminator
torch.manual_seed(0)
for kw in (dict(loss="wasserstein", clip_d=5.0, clip_g=0.5, diversity_weight=0.1, use_graph=True),
           dict(loss="hinge", use_graph=True), dict(loss="ns", use_graph=False, fuse_real_fake=False)):
    cfg = Config(embeddings_dimension=384, classes_count=1, batch_size=256)
    D = ViTDiscriminator(cfg).cuda().train(); G = SirenGenerator(fourier_features=True).cuda().train()
    eng = GanEngine(D, G, batch=256, **kw)
    for _ in range(6):
        l = eng.step(torch.rand(256, 3, 32, 32, device="cuda") * 2 - 1)
    torch.cuda.synchronize()
    assert torch.isfinite(l).all(), (kw, l)
    print(kw, [round(x, 4) for x in l.tolist()], "div", float(eng.div_loss), "norms", eng.clip_scratch[:, 0].tolist())
print("extra ok")
