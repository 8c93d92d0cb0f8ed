import torch
class Compose:
    def __init__(self, ts): self.ts = ts
    def __call__(self, x):
        for t in self.ts: x = t(x)
        return x
class Normalize:
    def __init__(self, mean, std): self.mean, self.std = mean, std
    def __call__(self, x):
        m = torch.as_tensor(self.mean, dtype=x.dtype, device=x.device).view(-1, 1, 1)
        s = torch.as_tensor(self.std, dtype=x.dtype, device=x.device).view(-1, 1, 1)
        return (x - m) / s
