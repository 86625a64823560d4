import torch
for p in (-2, -1, 0, 1, 2):
    try:
        s = torch.cuda.Stream(priority=p)
        print("priority", p, "->", s.priority)
    except Exception as e:
        print("priority", p, "error", str(e)[:100])
print(torch.cuda.current_stream().priority)
