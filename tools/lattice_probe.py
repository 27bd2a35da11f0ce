import sys, time; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from raytracer_amd import api, synth
lat = tuple(int(x) for x in sys.argv[1].split(','))
_, sc = synth.make("c3", lattice=lat)
ren = api.Renderer(0); ren.upload(sc)
rad = torch.empty((sc.height, sc.width*3), dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream()
for _ in range(3): ren.render_rows_device(rad.data_ptr(), 8, 0, 1, stream=st.cuda_stream); ren.finish()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(10): ren.render_rows_device(rad.data_ptr(), 8, 0, 1, stream=st.cuda_stream)
torch.cuda.synchronize(); print(sys.argv[1], "ms", round((time.perf_counter()-t0)*100,3), ren.finish())
