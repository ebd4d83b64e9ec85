/* envs.c — oracle restatement of CartPole, Chain and the step-limit wrappers.
 * TEST INFRASTRUCTURE (see oracle.h).  Built with -ffp-contract=off: Rust never fuses a*b+c.
 */
#include "oracle.h"
#include "../include/rl_detmath.h"

#include <math.h>

/* PhysicalConstants::default + EnvironmentParams::default (src/envs/cartpole.rs:178-216) */
void oracle_cartpole_default(oracle_cartpole *env) {
  env->gravity = 9.8;
  env->mass_cart = 1.0;
  env->mass_pole = 0.1;
  env->length_half_pole = 0.5;
  env->friction_cart = 0.01;
  env->friction_pole = 0.01;
  env->time_step = 0.02;
  env->action_force = 10.0;
  env->max_pos = 2.4;
  /* 12.0f64.to_radians(): Rust computes `self * (PI / 180.0)` */
  env->max_angle = 12.0 * (3.14159265358979323846 / 180.0);
  env->discount_factor = 0.99;
  env->use_libm = 0;
  oracle_cartpole_finish(env);
}

/* From<PhysicalConstants> for InternalPhysicalConstants (cartpole.rs:238-251) */
void oracle_cartpole_finish(oracle_cartpole *env) {
  double total_mass = env->mass_cart + env->mass_pole;
  env->total_weight = env->gravity * total_mass;
  env->inv_total_mass = 1.0 / total_mass;
  env->mass_length_pole = env->mass_pole * env->length_half_pole;
}

/* CartPole::initial_state (cartpole.rs:103-115): 4 draws of Uniform::new_inclusive(-0.05, 0.05)
 * in the order position, velocity, angle, angular velocity. */
void oracle_cartpole_initial_state(const oracle_cartpole *env, oracle_prng *rng, oracle_cartpole_state *s) {
  (void)env;
  s->x = oracle_prng_uniform_f64_inclusive(rng, -0.05, 0.05);
  s->xdot = oracle_prng_uniform_f64_inclusive(rng, -0.05, 0.05);
  s->th = oracle_prng_uniform_f64_inclusive(rng, -0.05, 0.05);
  s->thdot = oracle_prng_uniform_f64_inclusive(rng, -0.05, 0.05);
  s->nv_pos = 1;
}

/* InternalPhysicalConstants::angular_acceleration (cartpole.rs:398-431) */
static double angular_acceleration(const oracle_cartpole *c, double thdot, double applied_force,
                                   double signed_cart_friction, double w2, double sin_a, double cos_a) {
  double alpha = (-applied_force - c->mass_length_pole * w2 * (sin_a + signed_cart_friction * cos_a)) *
                 c->inv_total_mass;
  double beta = c->friction_pole * thdot / c->mass_length_pole;
  double numerator = c->gravity * sin_a + cos_a * (alpha + c->gravity * signed_cart_friction) - beta;
  double denominator =
      c->length_half_pole *
      (4.0 / 3.0 - c->mass_pole * cos_a * c->inv_total_mass * (cos_a - signed_cart_friction));
  return numerator / denominator;
}

/* InternalPhysicalConstants::normal_force (cartpole.rs:436-446) */
static double normal_force(const oracle_cartpole *c, double acc, double w2, double sin_a, double cos_a) {
  return c->total_weight - c->mass_length_pole * (acc * sin_a + w2 * cos_a);
}

/* InternalPhysicalConstants::next_state (cartpole.rs:306-387) */
void oracle_cartpole_next_state(const oracle_cartpole *c, const oracle_cartpole_state *s, double applied_force,
                                oracle_cartpole_state *out) {
  double signed_cart_friction = s->nv_pos ? c->friction_cart : -c->friction_cart;
  double sin_a, cos_a;
  if (c->use_libm) {
    sin_a = sin(s->th);
    cos_a = cos(s->th);
  } else {
    rl_sincos(s->th, &sin_a, &cos_a);
  }
  double w2 = s->thdot * s->thdot;
  double acc = angular_acceleration(c, s->thdot, applied_force, signed_cart_friction, w2, sin_a, cos_a);
  double nf = normal_force(c, acc, w2, sin_a, cos_a);
  /* f64::is_sign_positive: sign bit clear (so +0.0 and +NaN count as positive) */
  int nv_pos = !(rl_f64_bits(nf * s->xdot) >> 63);
  if (nv_pos != (s->nv_pos != 0)) {
    signed_cart_friction = -signed_cart_friction;
    acc = angular_acceleration(c, s->thdot, applied_force, signed_cart_friction, w2, sin_a, cos_a);
    nf = normal_force(c, acc, w2, sin_a, cos_a);
  }
  double force_pole = c->mass_length_pole * (w2 * sin_a + acc * cos_a);
  double force_friction = -signed_cart_friction * nf;
  double net_force = applied_force + force_pole + force_friction;
  double cart_acc = net_force * c->inv_total_mass;
  /* semi-implicit Euler (cartpole.rs:372-376) */
  double xdot = s->xdot + c->time_step * cart_acc;
  double x = s->x + c->time_step * xdot;
  double thdot = s->thdot + c->time_step * acc;
  double th = s->th + c->time_step * s->thdot;
  out->x = x;
  out->xdot = xdot;
  out->th = th;
  out->thdot = thdot;
  out->nv_pos = nv_pos;
}

/* CartPole::step (cartpole.rs:128-154) */
int oracle_cartpole_step(const oracle_cartpole *env, oracle_cartpole_state *s, int action, double *reward) {
  double applied_force = action == 0 ? -env->action_force : env->action_force; /* Push::Left = 0 */
  oracle_cartpole_state next;
  oracle_cartpole_next_state(env, s, applied_force, &next);
  *reward = 1.0;
  int terminal = fabs(next.x) > env->max_pos || fabs(next.th) > env->max_angle;
  if (terminal) return ORACLE_TERMINATE;
  *s = next;
  return ORACLE_CONTINUE;
}

/* Chain::default / Chain::step (src/envs/chain.rs:38-45, 83-105) */
void oracle_chain_default(oracle_chain *env) {
  env->size = 5;
  env->discount_factor = 0.95;
}

int oracle_chain_step(const oracle_chain *env, uint64_t *state, int action, oracle_prng *rng, double *reward) {
  return oracle_chain_step_draw(env, state, action, oracle_prng_gen_f32(rng), reward);
}

/* the same step with the slip draw `rng.gen::<f32>()` handed in (golden transition tables use recorded draws) */
int oracle_chain_step_draw(const oracle_chain *env, uint64_t *state, int action, float draw, double *reward) {
  if (draw < 0.2f) action = !action; /* Move::invert */
  if (action == 0) { /* Move::Left */
    *state = 0;
    *reward = 2.0;
  } else if (*state == env->size - 1) {
    *reward = 10.0;
  } else {
    *state += 1;
    *reward = 0.0;
  }
  return ORACLE_CONTINUE;
}

/* MemoryGame::default / initial_state / step (src/envs/memory.rs:33-40, 87-114) */
void oracle_memory_default(oracle_memory *env) {
  env->num_actions = 2;
  env->history_len = 1;
  env->discount_factor = 1.0;
}

void oracle_memory_initial_state(const oracle_memory *env, oracle_prng *rng, uint64_t *current, uint64_t *initial) {
  uint64_t state = oracle_prng_gen_range_u64(rng, 0, env->num_actions); /* rng.gen_range(0..num_actions) */
  *current = state;
  *initial = state;
}

int oracle_memory_step(const oracle_memory *env, uint64_t *current, uint64_t initial, uint64_t action,
                       double *reward) {
  if (*current == env->num_actions + env->history_len - 1) {
    *reward = action == initial ? 1.0 : -1.0;
    return ORACLE_TERMINATE;
  }
  *current = *current < env->num_actions ? env->num_actions : *current + 1;
  *reward = 0.0;
  return ORACLE_CONTINUE;
}

/* Wrapped<E, {Latent,Visible}StepLimit>::step tail (wrappers/step_limit.rs:82-88, 216-222) */
int oracle_step_limit_apply(int inner_successor, uint64_t *steps_remaining) {
  if (inner_successor == ORACLE_TERMINATE) return ORACLE_TERMINATE;
  *steps_remaining -= 1;
  if (inner_successor == ORACLE_CONTINUE && *steps_remaining == 0) return ORACLE_INTERRUPT;
  return inner_successor;
}

/* VisibleStepLimit::observe (step_limit.rs:194-200) */
double oracle_step_limit_remaining(uint64_t steps_remaining, uint64_t max_steps) {
  return (double)steps_remaining / (double)max_steps;
}

/* features_out of the derived product space: fields in declaration order, each `as f32`
 * (spaces/interval.rs:108-116; relearn_derive/src/space.rs:504-513; step_limit.rs:127-140) */
void oracle_cartpole_features(const oracle_cartpole_state *s, int limit_kind, uint64_t steps_remaining,
                              uint64_t max_steps, float *out) {
  out[0] = (float)s->x;
  out[1] = (float)s->xdot;
  out[2] = (float)s->th;
  out[3] = (float)s->thdot;
  if (limit_kind == ORACLE_LIMIT_VISIBLE) out[4] = (float)oracle_step_limit_remaining(steps_remaining, max_steps);
}

/* IndexSpace one-hot features (spaces/index.rs:97-116) */
void oracle_index_features(uint64_t index, uint64_t size, float *out) {
  for (uint64_t i = 0; i < size; ++i) out[i] = 0.0f;
  out[index] = 1.0f;
}
