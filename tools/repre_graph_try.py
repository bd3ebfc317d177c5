import sys, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import nsgp_repre_amd as N
dev = torch.device("cuda:0")
r = bench.repre_step(N, dev, 150, [0, 15, 20])
print(json.dumps({k: r[k] for k in ("repre_step_ms", "host_issue_ms", "graph_replay_ms", "isolated_pass_ms", "module_path_fp32_ms")}))
