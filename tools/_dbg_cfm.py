import sys, torch
sys.path.insert(0, "/root/repo")
from chatterbox_vllm_amd import cfm_weights as CW
from chatterbox_vllm_amd.cfm_decoder import CausalConditionalCFM
cfm = CausalConditionalCFM(CW.synthetic_state_dict(), max_frames=700)
L = 64
inp = CW.synthetic_inputs(L, 10)
ts = torch.tensor([0.0, 0.25])
x1 = cfm.solve_euler(inp["noise"], ts, inp["mu"], inp["spks"], inp["cond"])
z = inp["noise"].cuda(); zero = torch.zeros_like(z)
args = (torch.cat([z, z]), torch.cat([inp["mu"].cuda(), zero]), 0.0, torch.cat([inp["spks"].cuda(), torch.zeros(1, 80).cuda()]), torch.cat([inp["cond"].cuda(), zero]))
d = cfm.estimator(*args); d2 = cfm.estimator(*args)
print("estimator repeat equal", torch.equal(d, d2))
g = (1.0 + 0.7) * d[0:1] - 0.7 * d[1:2]
want = z + 0.25 * g
print("diff", (x1 - want).abs().max().item(), "n diff", (x1 != want).sum().item(), "of", x1.numel())
g2 = torch.tensor(1.7, dtype=torch.float32).cuda() * d[0:1] - torch.tensor(0.7, dtype=torch.float32).cuda() * d[1:2]
print("tensor-scalar form equal to python-scalar form", torch.equal(g, g2), torch.equal(z + 0.25 * g2, x1))
g3 = (d[0:1].double() * float(torch.tensor(1.7, dtype=torch.float32))).float() - (d[1:2].double() * float(torch.tensor(0.7, dtype=torch.float32))).float()
print("g3", torch.equal(g3, g), torch.equal(z + 0.25 * g3, x1))
