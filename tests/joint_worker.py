"""Worker of tests/test_gpu_joint_groups.py: ticks worlds with joints under the SFX_JOINT_GROUPS setting of its environment (read
once per process by libsfx) and prints a hash of every world's final state and the launches of its last tick."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DT = 1.0 / 60.0
MULTI_LAUNCH = 1


def main():
    from starframe_amd import scenes
    from starframe_amd.world import World

    out = {"per_world": []}

    def run(scene, n_ticks, **kw):
        w = World.from_scene(scene, flags=MULTI_LAUNCH, **kw)   # the launch-per-stage path, never the one-workgroup path
        for _ in range(n_ticks):
            w.step(DT)
        st = w.stats()
        assert st.overflow == 0, (scene.name, st.overflow)
        h = hashlib.sha256()
        pos, rot = w.poses()
        vel, om = w.velocities()
        for a in (pos, rot, vel, om):
            h.update(np.ascontiguousarray(a, dtype=np.float32).tobytes())
        out["per_world"].append([scene.name, int(st.n_joint_colors), int(st.launches), h.hexdigest()[:16]])
        w.close()

    run(scenes.chains(n_chains=40, links=30), 30)                      # chains: 30 joints a component
    run(scenes.chains(n_chains=6, links=200), 30)                      # 200 joints a component: one component a group
    run(scenes.chains(n_chains=3, links=400), 10)                      # larger than a group: the launch per colour
    for seed in range(4):                                              # soups: distance links, springs with damping, ropes, pins, hinges, angle locks
        run(scenes.random_soup(400 + 150 * seed, seed=900 + seed, extent=9.0, joints=0.12), 8)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
