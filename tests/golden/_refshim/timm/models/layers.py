import torch
class DropPath(torch.nn.Module):
    def __init__(self, p=0.0):
        super().__init__(); self.p = p
    def forward(self, x):
        assert self.p == 0.0 or not self.training
        return x
def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
    return torch.nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)
