#!/usr/bin/env python3
"""The reference's matlab/WAMPlannerExample.m with gpmp2_amd: 7-DOF WAM arm in the desk scene, signed distance
field built on the GPU, batch trajectory optimisation, dense up-sampling, collision cost, then one replanning
step (matlab/WAMReplannerExample.m:102-126).  Runs on an MI355X; there is no CPU path."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpmp2_amd as g

# ---- scene: occupancy grid -> signed distance field (both on the device)
dataset = g.generate3Ddataset("WAMDeskDataset")
t0 = time.perf_counter()
field = g.signedDistanceField3D(dataset.map, dataset.cell_size)          # [x][y][z] like the MATLAB utility
print(f"signed distance field {field.shape}: {1e3 * (time.perf_counter() - t0):.0f} ms")
layers = g.sdf3_zyx(field)                                               # field(:,:,z)' of the MATLAB script
sdf = g.SignedDistanceField([dataset.origin_x, dataset.origin_y, dataset.origin_z], dataset.cell_size,
                            layers.shape[1], layers.shape[2], layers.shape[0])
for z in range(layers.shape[0]):
    sdf.initFieldData(z, layers[z])

# ---- robot and settings (WAMPlannerExample.m:34-75)
arm = g.generateArm("WAMArm")
start_conf = np.array([-0.8, -1.70, 1.64, 1.29, 1.1, -0.106, 2.2])
end_conf = np.array([-0.0, 0.94, 0, 1.6, 0, -0.919, 1.55])
zero = np.zeros(7)
total_time_sec, total_time_step, total_check_step = 2.0, 10, 100
opt_setting = g.TrajOptimizerSetting(7)
opt_setting.set_total_step(total_time_step)
opt_setting.set_total_time(total_time_sec)
opt_setting.set_epsilon(0.2)
opt_setting.set_cost_sigma(0.02)
opt_setting.set_obs_check_inter(total_check_step // total_time_step - 1)
opt_setting.set_conf_prior_model(0.0001)
opt_setting.set_vel_prior_model(0.0001)
opt_setting.set_Qc_model(np.eye(7))
opt_setting.setDogleg()

# ---- batch plan
init_values = g.values_from_traj(g.initArmTrajStraightLine(start_conf, end_conf, total_time_step))
t0 = time.perf_counter()
result = g.BatchTrajOptimize3DArm(arm, sdf, start_conf, zero, end_conf, zero, init_values, opt_setting)
print(f"BatchTrajOptimize3DArm: {1e3 * (time.perf_counter() - t0):.1f} ms, "
      f"collision cost {g.CollisionCost3DArm(arm, sdf, result, opt_setting):.4f}")
dense = g.interpolateArmTraj(result, opt_setting.Qc, total_time_sec / total_time_step, 9)
print(f"up-sampled to {len(dense) // 2} states; x_50 = {np.round(dense[('x', 50)], 3)}")

# ---- replanning: execute to state 5, the goal moves (WAMReplannerExample.m:102-126)
isam = g.ISAM2TrajOptimizer3DArm(arm, sdf, opt_setting)
isam.initFactorGraph(start_conf, zero, end_conf, zero)
isam.initValues(result)
isam.update()
values = isam.values()
isam.fixConfigAndVel(5, values[("x", 5)], values[("v", 5)])
isam.changeGoalConfigAndVel(np.array([-0.6, 0.94, 0, 1.6, 0, -0.919, 1.55]), zero)
isam.update()
isam.update()
replanned = isam.values()
print(f"replanned: goal reached {np.round(replanned[('x', total_time_step)], 3)}, "
      f"state 5 moved by {np.abs(replanned[('x', 5)] - values[('x', 5)]).max():.1e}")
