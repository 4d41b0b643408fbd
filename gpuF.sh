cd $GRAFT_REPO_ROOT
for e in DartHumanWalker-v1 DartWalker3d-v1 DartHalfCheetah-v1; do
python - "$e" <<'PY'
import sys, numpy as np, time
import torch; torch.cuda.init()
from dart_env_amd import stepper as st
from dart_env_amd.model_card import card_for
env_id=sys.argv[1]
card=card_for(env_id); n=16384
s=st.HipStepper(card,n,precision=32); s.configure(st.CFG_AUTORESET,1); s.configure(st.CFG_STATS,1); s.configure(st.CFG_EPISODE_STATS,1)
s.reset(None,None,None,want_obs=False)
g=torch.Generator(device="cuda"); g.manual_seed(0)
ring=(torch.rand((16,n,card.act_dim),device="cuda",generator=g)*2-1).contiguous()
for t in range(600): s.step_device(ring[t%16].data_ptr())
s.sync()
q,dq=s.get_state(); h1,h2=s.solver_stats(); r,l,tot=s.episode_stats()
print(env_id,"finite",bool(np.isfinite(q).all() and np.isfinite(dq).all()),"max|dq|",float(np.abs(dq).max()),"fallbacks",int(h2[0]),"of",int(h2[1]),"solves; episodes",int(tot[2]),"mean len",tot[1]/max(tot[2],1),"mean return",tot[0]/max(tot[2],1))
PY
done
