import time, operator, torch
from transformers import BertConfig, BertModel
m = BertModel(BertConfig(hidden_size=128, num_hidden_layers=24, num_attention_heads=2, intermediate_size=256, vocab_size=1000), add_pooling_layer=False)
for dev in ("cpu", "cuda"):
    m = m.to(dev)
    plist = [p for n, p in m.named_parameters()]
    pairs = [(n, p) for n, p in m.named_parameters()]
    def t(f, n=300):
        f(); t0 = time.perf_counter()
        for _ in range(n): f()
        return (time.perf_counter() - t0) / n * 1e6
    vget = torch.Tensor._version.__get__
    ag = operator.attrgetter("_version")
    print(dev, len(plist), "params:",
          "walk %.0f us |" % t(lambda: list(m.named_parameters()), 30),
          "gen over pairs %.0f |" % t(lambda: tuple(p._version for _, p in pairs)),
          "gen over list %.0f |" % t(lambda: tuple(p._version for p in plist)),
          "map attrgetter %.0f |" % t(lambda: tuple(map(ag, plist))),
          "map descriptor %.0f |" % t(lambda: tuple(map(vget, plist))),
          "sum map %.0f |" % t(lambda: sum(map(vget, plist))),
          "data_ptr %.0f" % t(lambda: tuple(p.data_ptr() for p in plist)))
