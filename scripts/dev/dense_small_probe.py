import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from starframe_amd import scenes
from starframe_amd.world import World
DT = 1/60
for n, ext in ((500, 2.8), (500, 2.2), (300, 1.6)):
    s = scenes.random_soup(n, seed=12, extent=ext)
    for kw in (dict(), dict(tile_cells=2)):
        w = World.from_scene(s, **kw)
        try:
            for _ in range(50): w.step(DT)
            w.sync(); w.timer_begin()
            for _ in range(100): w.step(DT)
            ms = w.timer_end()/100
            st = w.stats()
            print(n, ext, kw, "ms %.4f pairs %d fallback %d colours %d overflow %d" % (ms, st.n_pairs, st.n_fallback, st.n_colors, st.overflow))
        except Exception as e:
            print(n, ext, kw, "ERR", str(e)[:80])
        w.close()
