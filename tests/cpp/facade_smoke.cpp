// Builds against include/gpmp2mi_planner.hpp with plain g++ (no HIP, no GTSAM) and links the
// product library.  On a box without a GPU the planner call must throw (no silent fallback);
// on a GPU box it optimises a tiny 2-link problem and prints the iteration count.
#include <cmath>
#include <cstdio>

#include "gpmp2mi_planner.hpp"

using namespace gpmp2mi;

int main() {
  try {
    Arm arm(2, {1.0, 1.0}, {0.0, 0.0}, {0.0, 0.0}, Pose3::Translation(0.0, 0.0, 0.0));
    BodySphereVector spheres;
    for (int l = 0; l < 2; l++)
      for (double x : {-0.75, -0.25}) spheres.emplace_back(l, 0.1, std::array<double, 3>{x, 0.0, 0.0});
    ArmModel model(arm, spheres);
    const int n = 60;
    Vector field(n * n);  // distance to a disc of radius 0.4 at (1.2, 1.0); column-major (row = y, col = x)
    for (int x = 0; x < n; x++)
      for (int y = 0; y < n; y++)
        field[x * n + y] = std::hypot(-3.0 + 0.1 * x - 1.2, -3.0 + 0.1 * y - 1.0) - 0.4;
    PlanarSDF sdf({-3.0, -3.0}, 0.1, n, n, field);
    TrajOptimizerSetting setting(2);
    setting.set_total_step(10);
    setting.set_total_time(2.0);
    setting.set_obs_check_inter(2);
    setting.set_cost_sigma(0.1);
    setting.set_epsilon(0.2);
    setting.setGaussNewton();
    const Vector start{0.0, 0.0}, end{1.5, 0.5}, zero{0.0, 0.0};
    const Trajectory init = initArmTrajStraightLine(start, end, 10);
    int iters = 0;
    double err = 0;
    const Trajectory out = BatchTrajOptimize2DArm(model, sdf, start, zero, end, zero, init, setting, &iters, &err);
    std::printf("OK iterations=%d final_error=%.6f x_5=(%.4f, %.4f) collision=%.4f\n", iters, err, out.x(5)[0],
                out.x(5)[1], CollisionCost2DArm(model, sdf, out, setting));
    const Trajectory dense = interpolateArmTraj(out, {}, 0.2, 4);
    if (dense.total_step != 50 || std::fabs(dense.x(50)[0] - out.x(10)[0]) > 0 || std::fabs(dense.x(5)[1] - out.x(1)[1]) > 0) return 5;
    std::printf("DENSE states=%zu x_27=(%.4f, %.4f)\n", dense.total_step + 1, dense.x(27)[0], dense.x(27)[1]);
    // factor classes: obstacle factor + numerical Jacobian, goal factor
    ObstaclePlanarSDFFactorArm of(0, model, sdf, 0.1, 0.2);
    const Vector q{1.0, 0.35};
    Vector H;
    const Vector e0 = of.evaluateError(q, &H);
    double worst = 0.0;
    for (int k = 0; k < 2; k++) {
      Vector qp = q, qm = q;
      qp[k] += 1e-6;
      qm[k] -= 1e-6;
      const Vector ep = of.evaluateError(qp), em = of.evaluateError(qm);
      for (std::size_t sidx = 0; sidx < e0.size(); sidx++)
        worst = std::fmax(worst, std::fabs((ep[sidx] - em[sidx]) / 2e-6 - H[sidx * 2 + k]));
    }
    GoalFactorArm gf(0, model, {2.0, 0.0, 0.0});
    const Vector ge = gf.evaluateError({0.0, 0.0});
    std::printf("FACTORS spheres=%zu max|H - Hnum|=%.1e goal_err=(%.1e, %.1e, %.1e)\n", e0.size(), worst, ge[0], ge[1], ge[2]);
    if (worst > 1e-5 || std::fabs(ge[0]) + std::fabs(ge[1]) + std::fabs(ge[2]) > 1e-12) return 6;
    // mobile manipulator through the same planner entry point
    Pose2MobileArmModel mmodel(Pose2MobileArm(arm, Pose3::Translation(0.1, 0.0, 0.0)),
                               {BodySphere(0, 0.3, {0.0, 0.0, 0.0}), BodySphere(1, 0.1, {-0.5, 0.0, 0.0}), BodySphere(2, 0.1, {-0.5, 0.0, 0.0})});
    TrajOptimizerSetting msetting(5);
    msetting.set_total_step(10);
    msetting.set_total_time(2.0);
    msetting.set_obs_check_inter(1);
    msetting.setLM();
    const Vector ms{-2.0, -2.0, 0.0, 0.0, 0.0}, me{-1.0, -2.2, 0.3, 0.5, 0.2}, mz(5, 0.0);
    int miters = 0;
    const Trajectory mout = BatchTrajOptimizePose2MobileArm2D(mmodel, sdf, ms, mz, me, mz, initArmTrajStraightLine(ms, me, 10),
                                                              msetting, &miters);
    std::printf("MOBILE iterations=%d x_10=(%.3f, %.3f, %.3f)\n", miters, mout.x(10)[0], mout.x(10)[1], mout.x(10)[2]);
    if (std::fabs(mout.x(10)[0] - me[0]) > 1e-2) return 7;
    // replanner: batch answer as initial values, fix state 3 where it is, move the goal, two updates
    ISAM2TrajOptimizer2DArm isam(model, sdf, setting);
    isam.initFactorGraph(start, zero, end, zero);
    isam.initValues(out);
    isam.update();
    const Vector fix_c{isam.values().x(3)[0], isam.values().x(3)[1]}, fix_v{isam.values().v(3)[0], isam.values().v(3)[1]};
    const Vector goal2{1.2, 0.9};
    isam.fixConfigAndVel(3, fix_c, fix_v);
    isam.changeGoalConfigAndVel(goal2, zero);
    isam.update();
    isam.update();
    const Trajectory& re = isam.values();
    const double dfix = std::hypot(re.x(3)[0] - fix_c[0], re.x(3)[1] - fix_c[1]);
    const double dgoal = std::hypot(re.x(10)[0] - goal2[0], re.x(10)[1] - goal2[1]);
    std::printf("REPLAN fixed_state_drift=%.2e goal_miss=%.2e\n", dfix, dgoal);
    if (dfix > 1e-3 || dgoal > 1e-3) return 4;
    return 0;
  } catch (const std::exception& e) {
    std::printf("EXCEPTION %s\n", e.what());
    return 3;
  }
}
