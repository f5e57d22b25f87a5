import os, sys, time, warnings, collections
import torch
sys.path.insert(0, os.getcwd())
import montecarlopredictivecoding_amd.utils.model as um
from montecarlopredictivecoding_amd.engine import Engine
from montecarlopredictivecoding_amd.utils.training_evaluation import get_mcpc_trainer, get_pc_trainer
warnings.simplefilter("ignore")
acc = collections.defaultdict(float)
def wrap(name):
    orig = getattr(Engine, name)
    def f(self, *a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = orig(self, *a, **k)
        torch.cuda.synchronize(); acc[name] += time.perf_counter() - t0
        return r
    setattr(Engine, name, f)
for n in ("bind_params", "bind_inputs", "bind_target", "load_state", "run", "store_state", "read_param_grads_flat", "store_adam_state", "sync_check"):
    wrap(n)
cfg = dict(input_size=30, hidden_size=256, hidden2_size=256, output_size=784, activation_fn="relu", T_pc=250, optimizer_x_fn_pc=torch.optim.Adam,
           optimizer_x_kwargs_pc={"lr": 0.1}, mixing=50, sampling=100, optimizer_x_kwargs_mcpc={"lr": 0.03}, optimizer_p_fn_mcpc=torch.optim.Adam,
           optimizer_p_kwargs_mcpc={"lr": 0.001}, loss_fn=um.bernoulli_fn, input_var=None)
B = int(os.environ.get("PB", "256"))
for dev in ("cuda:0", "cpu"):
    m = um.get_model(cfg, dev != "cpu")
    y = (torch.rand(B, 784) < 0.13).float().to(dev); inp = torch.zeros(B, 30, device=dev)
    pc_tr, mc_tr = get_pc_trainer(m, cfg, is_mcpc=True, training=False), get_mcpc_trainer(m, cfg, training=True)
    def it():
        pc_tr.train_on_batch(inputs=inp, loss_fn=um.bernoulli_fn, loss_fn_kwargs={"_target": y, "_var": None}, is_log_progress=False, is_return_results_every_t=False, is_checking_after_callback_after_t=False)
        mc_tr.train_on_batch(inputs=inp, loss_fn=um.bernoulli_fn, loss_fn_kwargs={"_target": y, "_var": None}, callback_after_t=um.random_step, callback_after_t_kwargs={"_pc_trainer": mc_tr}, is_sample_x_at_batch_start=False, is_log_progress=False, is_return_results_every_t=False, is_checking_after_callback_after_t=False)
    for _ in range(3): it()
    acc.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): it()
    torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / 5 * 1e3
    print(dev, "total %.2f ms/iter;" % tot, " ".join("%s %.2f" % (k, v / 5 * 1e3) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])))
