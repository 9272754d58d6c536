"""Which aten ops make up one no-grad UNet pass (development aid): op table + call sites of the strided copies / adds."""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["GD_GRAPHS"] = "0"
from geodiffuser_amd.diffusion import load_model
from geodiffuser_amd.attention_processors import VanillaAttentionProcessor
from torch.profiler import profile, ProfilerActivity
pipe, tok, sched = load_model(device="cuda:0", dtype=torch.bfloat16)
pipe.unet.set_attn_processor(VanillaAttentionProcessor())
B = int(os.environ.get("B", "3"))
x = torch.randn(B, 4, 64, 64, device="cuda", dtype=torch.bfloat16)
ctx = torch.randn(B, 77, 1024, device="cuda", dtype=torch.bfloat16)
with torch.no_grad():
    for _ in range(2):
        pipe.unet(x, 500, encoder_hidden_states=ctx)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        pipe.unet(x, 500, encoder_hidden_states=ctx)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=50, max_src_column_width=110))
