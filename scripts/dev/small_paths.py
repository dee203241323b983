"""Development: ms per tick of small worlds on the launch-per-stage path and on the one-workgroup path (same bits)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from starframe_amd import scenes, _abi
from starframe_amd.world import World

DT = 1.0 / 60.0
cases = {
    "pyramid (496 boxes, 10 substeps)": scenes.pyramid(),
    "soup 300": scenes.random_soup(300, seed=1, extent=4.0),
    "soup 150 dense": scenes.random_soup(150, seed=6, extent=2.0),
    "soup 500": scenes.random_soup(500, seed=7, extent=5.0),
    "soup 500 dense": scenes.random_soup(500, seed=8, extent=3.5),
    "soup 400 + joints + compounds": scenes.random_soup(400, seed=9, extent=5.0, joints=0.5, compounds=0.2),
    "circles 500": scenes.circles_container(n=500, width=30, floor_hw=5.0, wall_hh=8.0),
    "soup 1000": scenes.random_soup(1000, seed=2, extent=7.0),
    "soup 2000": scenes.random_soup(2000, seed=3, extent=10.0),
    "soup 600 + joints + compounds": scenes.random_soup(600, seed=4, extent=6.0, joints=0.5, compounds=0.2),
    "chains 8 x 60": scenes.chains(n_chains=8, links=60),
    "chains 20 x 100": scenes.chains(n_chains=20, links=100),
    "circles 1000": scenes.circles_container(n=1000, width=40, floor_hw=6.0, wall_hh=8.0),
    "circles 2000": scenes.circles_container(n=2000, width=60, floor_hw=8.0, wall_hh=8.0),
}
for name, s in cases.items():
    row = []
    for flags in (_abi.CFG_MULTI_LAUNCH, _abi.CFG_ONE_WORKGROUP):
        w = World.from_scene(s, flags=flags)
        for _ in range(150):
            w.step(DT)
        w.sync(); w.timer_begin()
        for _ in range(100):
            w.step(DT)
        ms = w.timer_end() / 100
        st = w.stats()
        row.append((ms, st.launches))
        w.close()
    print("%-34s bodies %5d pairs %6d colours %2d substeps %2d | stages %.4f ms (%3d launches)  one workgroup %.4f ms (%2d launches)  x%.2f" % (
        name, s.n, st.n_pairs, st.n_colors, s.substeps, row[0][0], row[0][1], row[1][0], row[1][1], row[0][0] / row[1][0]))
