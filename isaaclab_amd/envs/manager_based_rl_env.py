"""``ManagerBasedRLEnv.step`` call order over a synthetic physics feed
(``LAB/envs/manager_based_rl_env.py:152-241``; the MdpStepHarness of SURVEY.md section 8c).

The env reproduces the per-step order action -> decimation x {apply, write_data_to_sim, physics, update}
-> terminations -> rewards -> (reset) -> commands -> observations.  Physics is a
:class:`SyntheticPhysicsView` (direct writes or a replayed ring of frames); the reset path (events,
curriculum) is the reference's host-synchronising path and is optional.
"""
from __future__ import annotations

import os

import math
from typing import Any

import numpy as np
import torch

from ..assets import Articulation, SyntheticPhysicsView
from ..managers import (ActionManager, CommandManager, CurriculumManager, EventManager, ObservationManager,
                        RewardManager, TerminationManager)
from ..scene import InteractiveScene
from ..sensors import ContactSensor, RayCaster
from ..terrains import SyntheticTerrain, SyntheticTerrainCfg
from ..utils.warp import Mesh, convert_to_warp_mesh


class ManagerBasedRLEnv:
    def __init__(self, cfg, device: str = "cuda:0", physics: SyntheticPhysicsView | None = None,
                 mesh: Mesh | None = None, terrain: SyntheticTerrain | None = None, auto_reset: bool = False):
        cfg.validate() if hasattr(cfg, "validate") else None
        self.cfg = cfg
        self.device = device
        self.num_envs = cfg.scene.num_envs
        self.auto_reset = auto_reset
        self.fuse_post_physics = True   # episode counter + terminations + rewards in one launch
        self.fuse_substeps = True       # scene.update(k) + write_data_to_sim(k+1) in one launch
        self.fold_action = True         # process_action inside the first actuator launch of the step
        # last scene.update + terminations + rewards + command + observations + height-scan cast in one launch
        self.single_launch_post_physics = True
        # masked auto-reset: the stock event terms and the terrain curriculum ride in the reset branch of the post-physics
        # launch (False: one launch per term, as EventManager / CurriculumManager issue them)
        self.fold_events = os.environ.get("MDP_FOLD_EVENTS", "1") != "0"
        # ... and the two launches behind it (flagged recast of the reset envs, the reference's conditional actuator step)
        # are one when the actuator's gated launch can carry the recast (False: two launches, see concurrent_reset_tail)
        self.fuse_reset_tail = os.environ.get("MDP_FUSE_RESET_TAIL", "1") != "0"
        self.post_reset_noalloc = False  # masked resets inside that launch: use the allocation-free entry point
        # extras["log"] of a masked reset: 0 = a fresh copy every step (the reference's dict of new tensors); R > 0 = the
        # launches publish into a ring of R buffers and extras["log"] holds views of them, valid for R - 1 further steps
        # (no copy launch in the step: a training loop that reads its logs within R steps, e.g. once per rollout)
        self.reset_log_ring = 0
        self.concurrent_reset_tail = os.environ.get("MDP_CONCURRENT_RESET_TAIL", "1") != "0"
        # (terminations + rewards, command) and the height-scan cast on two streams (only without a reset in between)
        self.concurrent_post_physics = torch.device(device).type == "cuda"
        if self.concurrent_post_physics:
            self._side_stream = torch.cuda.Stream(device=device)
            self._ev_scene = torch.cuda.Event()
            self._ev_side = torch.cuda.Event()
        self.extras: dict[str, Any] = {}
        robot_cfg = cfg.scene.robot
        nj, nb = len(robot_cfg.joint_names), len(robot_cfg.body_names)
        self.physics = physics if physics is not None else SyntheticPhysicsView(self.num_envs, nj, nb, device)
        # -- scene
        env_origins = None
        if terrain is not None:
            env_origins = torch.tensor(terrain.env_origins, dtype=torch.float32, device=device)
        self.scene = InteractiveScene(cfg.scene, self.num_envs, device, env_origins, terrain)
        self.scene.articulations["robot"] = Articulation(robot_cfg, self.physics, device)
        robot = self.scene.articulations["robot"]
        if getattr(cfg.scene, "contact_forces", None) is not None:
            self.scene.sensors["contact_forces"] = ContactSensor(
                cfg.scene.contact_forces, self.physics, robot.body_names, self.num_envs, device, self.physics_dt)
        if getattr(cfg.scene, "height_scanner", None) is not None:
            if mesh is None:
                if terrain is None:
                    raise ValueError("the height scanner needs a mesh or a terrain")
                mesh = convert_to_warp_mesh(terrain.vertices, terrain.triangles, device)
            self.mesh = mesh
            rc = RayCaster(cfg.scene.height_scanner, self.physics, mesh, self.num_envs, device, self.physics_dt)
            rc.bind_pose(lambda d=robot.data: d.root_state_w)
            self.scene.sensors["height_scanner"] = rc
        # -- counters
        self.common_step_counter = 0
        self.episode_length_buf = torch.zeros(self.num_envs, device=device, dtype=torch.long)
        # -- managers (load order of manager_based_env.py / manager_based_rl_env.py:115-140)
        self.command_manager = CommandManager(cfg.commands, self)
        self.action_manager = ActionManager(cfg.actions, self)
        self.termination_manager = TerminationManager(cfg.terminations, self)
        self.observation_manager = ObservationManager(cfg.observations, self, seed=getattr(cfg, "seed", None))
        self.reward_manager = RewardManager(cfg.rewards, self)
        # reset / interval event terms (manager_based_env.py:131-133); tasks without an ``events`` cfg have none
        self.event_uniforms: dict[str, torch.Tensor] = {}     # tests: injected random numbers per event function
        self.event_manager = EventManager(getattr(cfg, "events", None), self)
        self.curriculum_manager = CurriculumManager(getattr(cfg, "curriculum", None), self)
        # the observation's consumer: see set_obs_normalizer
        self.obs_normalizer = None
        self.normalized_obs: torch.Tensor | None = None
        self._normalized_in_kernel = False
        self._sim_step_counter = 0
        self.obs_buf = None
        self.reward_buf = None
        self.reset_buf = None
        self.reset_terminated = None
        self.reset_time_outs = None

    # -- properties (manager_based_rl_env.py:93-102) ----------------------------------------------------
    @property
    def physics_dt(self) -> float:
        return self.cfg.sim.dt

    @property
    def step_dt(self) -> float:
        return self.cfg.sim.dt * self.cfg.decimation

    @property
    def max_episode_length_s(self) -> float:
        return self.cfg.episode_length_s

    @property
    def max_episode_length(self) -> int:
        return math.ceil(self.max_episode_length_s / self.step_dt)

    def set_obs_normalizer(self, scaler, out: torch.Tensor | None = None, group: str = "policy"):
        """Every step also leaves ``scaler(obs[group])`` -- standardised and clipped, evaluation mode -- in ``out``
        (``(num_envs, obs_dim)`` with any row stride, e.g. the current slot of the RL library's rollout storage; a buffer
        of the env's own when omitted), available as ``env.normalized_obs``.  On the single-launch path the kernels that
        produce the observation write it (``mdp_obs_normalize_t``), otherwise one ``mdp_scaler_apply`` follows the
        observation.  The running statistics are the caller's business (``scaler(x, train=True)``)."""
        self.obs_normalizer, self._normalize_group = scaler, group
        self.normalized_obs = out

    def _normalize_args(self, width: int):
        from .. import _lib as L

        sc = self.obs_normalizer
        if sc.size != width:
            raise ValueError(f"obs normalizer size {sc.size} != observation width {width}")
        if self.normalized_obs is None:
            self.normalized_obs = torch.empty((self.num_envs, width), dtype=torch.float32, device=self.device)
        out = self.normalized_obs
        if out.shape != (self.num_envs, width) or out.dtype != torch.float32 or out.stride(1) != 1:
            raise ValueError("normalized_obs must be a float32 (num_envs, obs_dim) tensor with unit column stride")
        a = L.MdpObsNormalize()
        a.mean32, a.denom, a.out = sc._mean32.data_ptr(), sc._denom.data_ptr(), out.data_ptr()
        a.ld_out, a.clip = out.stride(0), float(sc.clip_threshold)
        return a

    # -- stepping -----------------------------------------------------------------------------------------
    def step(self, action: torch.Tensor):
        """manager_based_rl_env.py:152-241"""
        # the first actuator launch of the step processes the action itself when the terms allow it
        deferred = self.action_manager.deferred_action(action.to(self.device)) if self.fold_action else None
        if deferred is None:
            self.action_manager.process_action(action.to(self.device))
        decimation = self.cfg.decimation
        masked = self.auto_reset == "masked"
        concurrent = (self.concurrent_post_physics and (not self.auto_reset or masked) and self._can_fuse_post_physics()
                      and not self._post_physics_reads_lazy_sensor())
        om = self.observation_manager
        # sensor-owned observation columns (height scan) come from the cast kernel, on its own stream
        split = concurrent and om.can_split_sensor_terms()
        early_cast = split and self.scene.fused_update and len(om.scan_sensors()) > 0
        # a task without ray casters: the same single launch, manager blocks only (the flat velocity tasks)
        flat_single = (concurrent and split and not early_cast and self.scene.fused_update and self.single_launch_post_physics
                       and not self._has_lazy_sensor() and os.environ.get("MDP_FLAT_SINGLE_LAUNCH") != "0")
        has_events = (any(m in self.event_manager.available_modes for m in ("reset", "interval"))
                      or len(self.curriculum_manager.active_terms) > 0)
        fold_events = None
        if has_events:
            # event terms write the robot state between rewards and observations.  The stock terms of the velocity tasks ride
            # in the reset branch of the post-physics launch (mdp_post_physics_step_with_reset_events); anything else: the
            # stages run one after the other
            fold_events = self._foldable_events() if (masked and self.fold_events) else None
            if fold_events is None:
                split = early_cast = concurrent = flat_single = False
        if masked:
            flat_single = False
        if concurrent and not split and not self._has_lazy_sensor() and os.environ.get("MDP_FORCE_CONCURRENT") != "1":
            # nothing to overlap (a task without ray casters): the second stream would only add its cross-queue
            # hand-over (~25 us on this stack) to a chain that is serial anyway
            concurrent = False
        if masked and not (early_cast and self.single_launch_post_physics):
            split = early_cast = concurrent = False   # resets sit between rewards and commands: serial launches below
        self.action_manager.apply_action()
        self.scene.write_data_to_sim(joint_action=deferred)
        for k in range(decimation):
            self._sim_step_counter += 1
            self.physics.advance()                      # sim.step(render=False): synthetic / replayed
            if k + 1 < decimation:
                # scene.update(k), apply_action(k+1), write_data_to_sim(k+1): nothing steps physics between them, so
                # they share one launch when the actuator can carry the update (same values, same order of writes)
                self.action_manager.apply_action()
                if self.fuse_substeps:
                    self.scene.update_and_write(self.physics_dt)
                else:
                    self.scene.update(dt=self.physics_dt)
                    self.scene.write_data_to_sim()
            elif not early_cast and not flat_single:
                self.scene.update(dt=self.physics_dt)
            # (with early_cast the last scene.update runs after the loop, next to or inside the post-physics launches)
        self.common_step_counter += 1
        if early_cast and self.single_launch_post_physics and self._post_physics_single_launch(reset=masked, events=fold_events):
            pass
        elif flat_single and self._post_physics_single_launch():
            pass
        elif flat_single:                               # the launch declined the task: the last update is still due
            self.scene.update(dt=self.physics_dt)
            self._termination_reward_fused()
            self.command_manager.compute(dt=self.step_dt)
            self.obs_buf = self.observation_manager.compute()
        elif concurrent and not masked:
            # Without the reset path between them (manager_based_rl_env.py:214-229) the chain terminations -> rewards ->
            # command update -> observations and the height-scan cast do not depend on each other: they go to two
            # streams (two branches of a captured hipGraph) and join at the end of the step.  The cast reads the sensor
            # pose from the physics transforms and does its own update(dt) bookkeeping, so it only waits for the last
            # physics step, not for the scene update (which moves to the other branch).
            main = torch.cuda.current_stream(self.device)
            side = self._side_stream
            self._ev_scene.record(main)
            with torch.cuda.stream(side):
                side.wait_event(self._ev_scene)
                if early_cast:
                    self.scene.update(dt=self.physics_dt, skip_ticks=om.scan_sensors())
                self._termination_reward_fused()
                self.command_manager.compute(dt=self.step_dt)   # after the rewards: they read the previous command
                if split:
                    self.obs_buf = om.compute(skip_sensor_terms=True)
                self._ev_side.record(side)
            if early_cast:
                om.compute_sensor_terms(tick_dt=self.physics_dt, from_physics=True)
            elif split:
                om.compute_sensor_terms()
            else:
                for sensor in self.scene.sensors.values():  # lazily evaluated sensors the observations will read
                    if hasattr(sensor, "fill_scene_tick"):
                        sensor._update_outdated_buffers()
            main.wait_event(self._ev_side)
            if not split:
                self.obs_buf = om.compute()
        else:
            if early_cast:                              # masked resets and the single launch declined the task
                self.scene.update(dt=self.physics_dt)
            if self.fuse_post_physics and self._can_fuse_post_physics():
                self._termination_reward_fused()
            else:
                self.episode_length_buf += 1
                self.reset_buf = self.termination_manager.compute()
                self.reward_buf = self.reward_manager.compute(dt=self.step_dt)
            if self.auto_reset == "masked":
                self._reset_masked(self.reset_buf)
            elif self.auto_reset:
                reset_env_ids = self.reset_buf.nonzero(as_tuple=False).squeeze(-1)
                if len(reset_env_ids) > 0:
                    self._reset_idx(reset_env_ids)
                    self.scene.write_data_to_sim()      # manager_based_rl_env.py:221 (one more actuator evaluation)
            self.command_manager.compute(dt=self.step_dt)
            if "interval" in self.event_manager.available_modes:
                self.event_manager.apply(mode="interval", dt=self.step_dt)      # manager_based_rl_env.py:233-235
            self.obs_buf = self.observation_manager.compute()
        if self.obs_normalizer is not None:
            if not self._normalized_in_kernel:
                obs = self.obs_buf[self._normalize_group]
                self._normalize_args(obs.shape[1])
                self.obs_normalizer(obs, out=self.normalized_obs)
            self._normalized_in_kernel = False
        self.reset_terminated = self.termination_manager.terminated
        self.reset_time_outs = self.termination_manager.time_outs
        return self.obs_buf, self.reward_buf, self.reset_terminated, self.reset_time_outs, self.extras

    def _has_lazy_sensor(self) -> bool:
        return any(hasattr(sensor, "fill_scene_tick") for sensor in self.scene.sensors.values())

    def _can_fuse_post_physics(self) -> bool:
        tm, rm = self.termination_manager, self.reward_manager
        # user-defined (non-lowered) terms may read the episode counter or the termination buffers from Python
        if tm._program is None or rm._program is None:
            return False            # first step: the unfused calls build the programs (and discover user terms)
        return (len(tm.active_terms) > 0 and len(rm.active_terms) > 0 and not tm._external and not rm._external)

    def _post_physics_reads_lazy_sensor(self) -> bool:
        """True when a termination / reward term reads a lazily evaluated sensor (e.g. ``base_height_l2`` with a ray
        caster): its cast is then ordered before that term and cannot overlap it."""
        for prog in (self.termination_manager._program, self.reward_manager._program):
            if prog is not None and any(getattr(spec, "lazy_sensors", None) for spec in prog.specs):
                return True
        return False

    def _foldable_events(self):
        """The task's reset / interval event terms and its curriculum as a plan for the reset branch of the post-physics
        launch, or None when it holds anything but the stock terms in the order that launch runs them: (curriculum
        ``terrain_levels_vel``,) ``reset_root_state_uniform``, ``reset_joints_by_scale`` / ``_by_offset``, and ONE interval
        term ``push_by_setting_velocity`` with per-env timers (velocity_env_cfg.py:187-218, 269-272)."""
        from . import mdp as M
        from ..tasks.velocity.mdp.curriculums import terrain_levels_vel

        em, cm = self.event_manager, self.curriculum_manager
        if any(m not in ("reset", "interval", "startup", "prestartup") for m in em.available_modes):
            return None
        if any(em._mode_class_term_cfgs.get(m) for m in ("reset", "interval")) or cm._class_term_cfgs:
            return None
        plan = {"root": None, "joints": None, "push": None, "cur": None, "mark": []}
        for i, cfg in enumerate(em._mode_term_cfgs.get("reset", [])):
            if cfg.min_step_count_between_reset != 0:
                return None
            if cfg.func is M.reset_root_state_uniform and plan["root"] is None and plan["joints"] is None:
                plan["root"] = cfg
            elif cfg.func is M.reset_joints_by_scale and plan["joints"] is None:
                plan["joints"] = (cfg, 0, "reset_joints_by_scale")
            elif cfg.func is M.reset_joints_by_offset and plan["joints"] is None:
                plan["joints"] = (cfg, 1, "reset_joints_by_offset")
            else:
                return None
            plan["mark"].append(i)
        if len(plan["mark"]) > 8:
            return None
        interval = em._mode_term_cfgs.get("interval", [])
        if len(interval) > 1:
            return None
        if interval:
            if interval[0].func is not M.push_by_setting_velocity or interval[0].is_global_time:
                return None
            plan["push"] = interval[0]
        if len(cm._term_cfgs) > 1:
            return None
        if cm._term_cfgs:
            if cm._term_cfgs[0].func is not terrain_levels_vel:
                return None
            plan["cur"] = (cm._term_names[0], cm._term_cfgs[0])
        return plan

    def _post_events_args(self, plan, keep: list):
        """``mdp_post_events_t`` of a ``_foldable_events`` plan (the term structs are appended to ``keep``)."""
        import ctypes as C

        from .. import _lib as L
        from ..managers.scene_entity_cfg import SceneEntityCfg
        from ..tasks.velocity.mdp.curriculums import _fill_terrain_levels_vel
        from .mdp.events import _fill_push, _fill_reset_joints, _fill_reset_root_state

        em = self.event_manager
        robot = SceneEntityCfg("robot")
        ev = L.MdpPostEvents()
        if plan["cur"] is not None:
            a, _ = _fill_terrain_levels_vel(self, None, plan["cur"][1].params.get("asset_cfg", robot), keep)
            keep.append(a)
            ev.curriculum = C.addressof(a)
        if plan["root"] is not None:
            pr = plan["root"].params
            a = _fill_reset_root_state(self, None, pr["pose_range"], pr["velocity_range"], pr.get("asset_cfg", robot), keep,
                                       body_acc=False)
            keep.append(a)
            ev.reset_root_state = C.addressof(a)
        if plan["joints"] is not None:
            cfg, mode, name = plan["joints"]
            pr = cfg.params
            a = _fill_reset_joints(self, None, pr["position_range"], pr["velocity_range"], pr.get("asset_cfg", robot), mode,
                                   name, keep)
            keep.append(a)
            ev.reset_joints = C.addressof(a)
        if plan["mark"]:
            a = L.MdpEventResetMark()
            a.num_terms, a.step_count = len(plan["mark"]), int(self._sim_step_counter // self.cfg.decimation)
            for k, i in enumerate(plan["mark"]):
                a.last_triggered_step[k] = em._reset_term_last_triggered_step_id[i].data_ptr()
                a.triggered_once[k] = em._reset_term_last_triggered_once[i].data_ptr()
            keep.append(a)
            ev.reset_mark = C.addressof(a)
        if plan["push"] is not None:
            cfg = plan["push"]
            pr = cfg.params
            a = _fill_push(self, None, pr["velocity_range"], pr.get("asset_cfg", robot), keep, body_acc=False)
            keep.append(a)
            ev.push = C.addressof(a)
            lower, upper = cfg.interval_range_s
            u = em.interval_uniforms.get(0)
            ev.push_time_left = em._interval_term_time_left[0].data_ptr()
            ev.push_lo, ev.push_scale = float(lower), float(upper - lower)
            ev.push_tick_u = u.data_ptr() if u is not None else None
            ev.push_tick_rng = None if u is not None else em._interval_rng[0].data_ptr()
            ev.push_trigger = em._interval_masks[0].data_ptr()
            ev.interval_dt = float(self.step_dt)
        return ev

    def _post_physics_single_launch(self, reset: bool = False, events=None) -> bool:
        """manager_based_rl_env.py:195-238 (last ``scene.update``, episode counter, terminations, rewards, command,
        observations incl. the height-scan cast) as ONE launch, ``mdp_post_physics_step``.  False when the task does not
        have the shape that entry point covers; the caller then uses the separate launches."""
        import ctypes as C

        from .. import _lib as L

        om, cm = self.observation_manager, self.command_manager
        oa = om.post_physics_args()
        if oa is None:
            return False
        obs_args, sensor, scan, oprog, group_name, out = oa
        terms = list(cm._terms.values())
        if len(terms) != 1 or not hasattr(terms[0], "fill_post_physics"):
            return False
        if sensor is not None and not (sensor._pattern_vertical_down and sensor.cfg.attach_yaw_only
                                       and self._mesh_has_grid(sensor._mesh)):
            return False
        if sensor is None and reset:
            return False
        plan = self.scene._scene_update_desc(self.physics_dt, skip_ticks=(sensor,) if sensor is not None else ())
        if plan is None:
            return False
        d, art = plan
        art.data.commit_scene_update(self.physics_dt)   # the lazy buffers are fresh once the launch below has run
        tm, rm = self.termination_manager, self.reward_manager
        tp, rp = tm.prepare_launch(), rm.prepare_launch()
        tr = L.MdpTerminationRewardArgs()
        tr.termination, tr.reward, tr.dt = tp.handle, rp.handle, float(self.step_dt)
        tr.episode_length_buf_inc = self.episode_length_buf.data_ptr()
        tr.term_dones, tr.truncated = tm._term_dones_block.data_ptr(), tm._truncated_buf.data_ptr()
        tr.terminated, tr.dones = tm._terminated_buf.data_ptr(), tm._dones_buf.data_ptr()
        tr.reward_buf, tr.episode_sums = rm._reward_buf.data_ptr(), rm._episode_sums_block.data_ptr()
        tr.step_reward = rm._step_reward.data_ptr()
        ca = terms[0].fill_post_physics(self.step_dt)
        if self.obs_normalizer is not None and self._normalize_group == group_name:
            na = self._normalize_args(out.shape[1])
            obs_args.normalize = na
            if scan is not None:
                scan.normalize = na
            self._normalized_in_kernel = True
        elif scan is not None:
            scan.normalize.out = None
        cast = sensor.fill_post_physics(scan, self.physics_dt) if sensor is not None else None
        oprog.run_before()
        if reset:
            # reset branch inside the launch; the ray caster's own reset follows it (see mdp_post_physics_step_with_reset)
            if getattr(self, "_post_reset_plan", None) is None:
                self._post_reset_plan = self._masked_reset_plan(separate_sensor=sensor)
            # statistics rows: count, reward terms, two command metrics, termination terms (+ the mean terrain level)
            rows = 1 + len(rm.active_terms) + 2 + len(tm.active_terms) + (1 if events and events["cur"] is not None else 0)
            if getattr(self, "_post_reset_rows", None) != rows:
                self._post_reset_rows = rows
                # block ticket (zeroed once) + one slot of row sums per manager block of the launch
                self._post_reset_sums = torch.zeros(1 + rows * ((self.num_envs + 7) // 8), dtype=torch.float64,
                                                    device=self.device)
                self._post_reset_log = torch.zeros(rows, dtype=torch.float32, device=self.device)
                self._post_reset_log_ring = None
            self._next_reset_log()
            _, count, _, blob = self._post_reset_plan
            ra = L.MdpPostResetArgs()
            ra.fills, ra.num_fills, ra.max_episode_length_s = blob.data_ptr(), count, float(self.max_episode_length_s)
            ra.log_sums, ra.log = self._post_reset_sums.data_ptr(), self._post_reset_log.data_ptr()
            launch = L.lib.mdp_post_physics_step_with_reset
            if events is not None:
                # curriculum + event terms inside the launch's reset branch (the order of _reset_idx / step)
                keep: list = []
                ev = self._post_events_args(events, keep)
                L.check(L.lib.mdp_post_physics_step_with_reset_events(
                    self.num_envs, C.byref(d), float(self.physics_dt), C.byref(tr), C.byref(ca), C.byref(obs_args),
                    C.byref(cast), C.byref(ra), C.byref(ev), L.stream(self.device)))
                from .mdp.events import _invalidate_body_state

                _invalidate_body_state(art.data)              # articulation.py:337-339, 503-505
                if events["root"] is not None:
                    art.data.defer_body_acc_zero(tm._dones_buf)
                if events["push"] is not None:
                    art.data.defer_body_acc_zero(self.event_manager._interval_masks[0])
                self.event_manager.reset(tm._dones_buf)
                launch = None
            elif self.post_reset_noalloc:
                # callers that must not allocate (graph capture, serving loops): the library's own static scratch array
                # takes the place of the workspace (mdp_post_physics_step_with_reset_noalloc; same kernels, same outputs)
                ra.log_sums = None
                launch = L.lib.mdp_post_physics_step_with_reset_noalloc
            if launch is not None:
                L.check(launch(self.num_envs, C.byref(d), float(self.physics_dt), C.byref(tr), C.byref(ca),
                               C.byref(obs_args), C.byref(cast), C.byref(ra), L.stream(self.device)))
            # RayCaster.reset of the done envs + their recast (the observation's lazy evaluation in the reference) and the
            # reference's post-reset write_data_to_sim (gate: log[0] = number of reset envs).  Neither reads what the
            # other writes (sensor buffers + scan columns / actuator state + joint efforts), and both are short chains of
            # dependent loads: on two streams (two branches of a captured hipGraph) they cost the longer one.
            arts = list(self.scene.articulations.values())
            if self.fuse_reset_tail and len(arts) == 1 and not self.scene.rigid_objects and arts[0].can_carry_recast():
                # ONE launch: recast blocks in front of the gated actuator blocks (mdp_actuator_lstm_compute_gated_with_recast)
                self.scene.write_data_to_sim(gate=self._post_reset_log, recast=(cast, tm._dones_buf))
            elif self.concurrent_reset_tail and self.concurrent_post_physics:
                main = torch.cuda.current_stream(self.device)
                side = self._side_stream
                self._ev_scene.record(main)
                with torch.cuda.stream(side):
                    side.wait_event(self._ev_scene)
                    L.check(L.lib.mdp_raycaster_update_flagged(self.num_envs, C.byref(cast), tm._dones_buf.data_ptr(),
                                                               L.stream(self.device)))
                    self._ev_side.record(side)
                self.scene.write_data_to_sim(gate=self._post_reset_log)
                main.wait_event(self._ev_side)
            else:
                L.check(L.lib.mdp_raycaster_update_flagged(self.num_envs, C.byref(cast), tm._dones_buf.data_ptr(),
                                                           L.stream(self.device)))
                self.scene.write_data_to_sim(gate=self._post_reset_log)
            self._publish_reset_log()
            if events is not None and events["cur"] is not None:
                # the term's state (curriculum_manager.py:119-134) and its log line (:88-117): the launch's last row
                name = events["cur"][0]
                level = self._post_reset_log_last[rows - 1]
                self.curriculum_manager._curriculum_state[name] = level
                self.extras["log"]["Curriculum/" + name] = level
        else:
            L.check(L.lib.mdp_post_physics_step(self.num_envs, C.byref(d), float(self.physics_dt), C.byref(tr), C.byref(ca),
                                                C.byref(obs_args), C.byref(cast) if cast is not None else None,
                                                L.stream(self.device)))
        self.reset_buf = tm._dones_buf
        self.reward_buf = rm._reward_buf
        om._obs_buffer = {group_name: out}
        self.obs_buf = om._obs_buffer
        self._single_launch_steps = getattr(self, "_single_launch_steps", 0) + 1
        return True

    def _next_reset_log(self):
        """With ``reset_log_ring`` = R > 0: point ``_post_reset_log`` at the next of R buffers (this step's log)."""
        R = int(self.reset_log_ring)
        if R <= 0:
            return
        ring = getattr(self, "_post_reset_log_ring", None)
        if ring is None or ring.shape[0] != R:
            ring = torch.zeros((R, self._post_reset_log.shape[0]), dtype=torch.float32, device=self.device)
            self._post_reset_log_ring, self._post_reset_log_slot = ring, 0
        self._post_reset_log = ring[self._post_reset_log_slot]
        self._post_reset_log_slot = (self._post_reset_log_slot + 1) % R

    def _publish_reset_log(self):
        """``extras["log"]`` of the step: views of a copy of the values the post-physics launch published (device
        tensors; reward_manager.py:115-121, command_manager.py:131-136, termination_manager.py:149-150)."""
        rm, tm, cm = self.reward_manager, self.termination_manager, self.command_manager
        # the launch overwrites its buffer next step and callers keep these: a copy, or this step's slot of the ring
        p = self._post_reset_log if int(self.reset_log_ring) > 0 else self._post_reset_log.clone()
        self._post_reset_log_last = p
        nr = len(rm.active_terms)
        log = {}
        for i, name in enumerate(rm.active_terms):
            log["Episode_Reward/" + name] = p[1 + i]
        term_name = next(iter(cm._terms))
        log[f"Metrics/{term_name}/error_vel_xy"] = p[1 + nr]
        log[f"Metrics/{term_name}/error_vel_yaw"] = p[2 + nr]
        for i, name in enumerate(tm.active_terms):
            log["Episode_Termination/" + name] = p[3 + nr + i]
        self.extras["log"] = log

    def _mesh_has_grid(self, mesh) -> bool:
        cache = self.__dict__.setdefault("_mesh_grid_cache", {})
        if id(mesh) not in cache:
            cache[id(mesh)] = mesh.info().get("accel") == "grid"
        return cache[id(mesh)]

    def _termination_reward_fused(self):
        """manager_based_rl_env.py:199-206: ``episode_length_buf += 1``, ``termination_manager.compute()`` and
        ``reward_manager.compute(dt)`` as one kernel launch (``mdp_termination_reward_compute``)."""
        from .. import _lib as L

        tm, rm = self.termination_manager, self.reward_manager
        tp = tm.prepare_launch()
        rp = rm.prepare_launch()
        L.check(L.lib.mdp_termination_reward_compute(
            tp.handle, rp.handle, float(self.step_dt), self.episode_length_buf.data_ptr(),
            tm._term_dones_block.data_ptr(), tm._truncated_buf.data_ptr(), tm._terminated_buf.data_ptr(),
            tm._dones_buf.data_ptr(), rm._reward_buf.data_ptr(), rm._episode_sums_block.data_ptr(),
            rm._step_reward.data_ptr(), L.stream(self.device)))
        self.reset_buf = tm._dones_buf
        self.reward_buf = rm._reward_buf

    # -- reset by mask (the reset path without its host syncs) ------------------------------
    def _masked_reset_plan(self, separate_sensor=None):
        """The per-env state ``_reset_idx`` zeroes, as ``mdp_fill_t`` records (built once).  Raises
        ``NotImplementedError`` for state this path does not know how to reset.  ``separate_sensor``: its three
        bookkeeping arrays stay out of the table (``mdp_raycaster_update_flagged`` resets them after the launch)."""
        import ctypes as C
        import struct

        from .. import _lib as L
        from ..envs.mdp.commands.velocity_command import UniformVelocityCommand

        fills, keep = [], []

        def add(t: torch.Tensor, count: int, env_stride: int, elem_stride: int = 1, value=0):
            f = L.MdpFill()
            f.ptr, f.elem_bytes, f.count = t.data_ptr(), t.element_size(), int(count)
            f.env_stride, f.elem_stride = int(env_stride), int(elem_stride)
            if t.dtype == torch.float32:
                f.value_bits = struct.unpack("<I", struct.pack("<f", float(value)))[0]
            else:
                f.value_bits = int(value)
            fills.append(f)
            keep.append(t)

        def add_rows(t: torch.Tensor, value=0):          # (N, ...) contiguous
            if not t.is_contiguous() or t.shape[0] != self.num_envs:
                raise NotImplementedError("masked reset needs contiguous (N, ...) buffers")
            per_env = t.numel() // max(self.num_envs, 1)
            add(t, per_env, per_env, 1, value)

        n = self.num_envs
        # scene.reset (articulation.py:163-167, sensor_base.py:182-195, contact_sensor.py:136-155, ray_caster.py:106-113)
        for art in self.scene.articulations.values():
            for act in art.actuators.values():
                if hasattr(act, "sea_hidden_state"):
                    layers, rows, hidden = act.sea_hidden_state.shape
                    per_env = (rows // n) * hidden
                    for state in (act.sea_hidden_state, act.sea_cell_state):
                        for l in range(layers):
                            add(state[l], per_env, per_env)
                elif type(act).__name__ in ("ImplicitActuator", "IdealPDActuator", "DCMotor"):
                    pass                                      # stateless
                else:
                    raise NotImplementedError(f"masked reset of actuator {type(act).__name__}")
        for sensor in self.scene.sensors.values():
            if sensor is not separate_sensor:
                add_rows(sensor._timestamp)
                add_rows(sensor._timestamp_last_update)
                add_rows(sensor._is_outdated, 1)
            if isinstance(sensor, ContactSensor):
                d = sensor.data
                add_rows(d.net_forces_w)
                if d.net_forces_w_history.data_ptr() != d.net_forces_w.data_ptr():
                    add_rows(d.net_forces_w_history)
                if getattr(d, "current_air_time", None) is not None:
                    for t in (d.current_air_time, d.last_air_time, d.current_contact_time, d.last_contact_time):
                        add_rows(t)
            elif isinstance(sensor, RayCaster):
                if tuple(sensor.cfg.drift_range) != (0.0, 0.0):
                    raise NotImplementedError("masked reset with a ray-caster drift range")
                add_rows(sensor.drift)
            else:
                raise NotImplementedError(f"masked reset of sensor {type(sensor).__name__}")
        # managers (manager_based_rl_env.py:365-391)
        om, am, rm, tm, cm = (self.observation_manager, self.action_manager, self.reward_manager,
                              self.termination_manager, self.command_manager)
        if getattr(om, "_group_obs_class_term_cfgs", None) and any(om._group_obs_class_term_cfgs.values()):
            raise NotImplementedError("masked reset of class-based observation terms")
        if any(getattr(c, "history_length", 0) for cfgs in om._group_obs_term_cfgs.values() for c in cfgs):
            raise NotImplementedError("masked reset of observation histories")
        if rm._class_term_cfgs or tm._class_term_cfgs:
            raise NotImplementedError("masked reset of class-based reward / termination terms")
        add_rows(am._action)
        add_rows(am._prev_action)
        for term in am._terms.values():
            if term.raw_actions.data_ptr() not in (am._action.data_ptr(),):
                add_rows(term.raw_actions)
        add(rm._episode_sums_block, len(rm.active_terms), 1, n)
        self._reset_cmd_terms = []
        for term in cm._terms.values():
            if not isinstance(term, UniformVelocityCommand):
                raise NotImplementedError(f"masked reset of command term {type(term).__name__}")
            for v in term.metrics.values():
                add_rows(v)
            add_rows(term.command_counter)
            self._reset_cmd_terms.append(term)
        add_rows(self.episode_length_buf)
        if len(fills) > 48:
            raise NotImplementedError("masked reset: more than MDP_MAX_FILLS buffers")
        arr = (L.MdpFill * len(fills))(*fills)
        if separate_sensor is None:
            return arr, len(fills), keep
        # device copy of the table for the post-physics launch (the separated sensor is reset by
        # mdp_raycaster_update_flagged, which needs no table)
        blob = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        return arr, len(fills), keep, blob

    def _reset_masked(self, mask: torch.Tensor):
        """``_reset_idx`` for the envs flagged in ``mask`` (``reset_buf``) with no host synchronisation: statistics as
        device tensors, every zeroed buffer in ONE launch (``mdp_masked_fill``), commands resampled by their masked
        kernel.  Events / curriculum stay out of scope like in ``_reset_idx``.  With an empty mask it is a no-op on the
        state; the logged means are then 0 (the reference does not log at all in such a step)."""
        from .. import _lib as L

        if getattr(self, "_reset_plan", None) is None:
            self._reset_plan = self._masked_reset_plan()
        arr, count, _ = self._reset_plan
        rm, tm, cm = self.reward_manager, self.termination_manager, self.command_manager
        # curriculum first (manager_based_rl_env.py:353): it reads the command and the root position of the episode
        self.curriculum_manager.sync_free = True
        self.curriculum_manager.compute(env_ids=mask)
        # statistics before anything is zeroed (reward_manager.py:115-121, command_manager.py:131-136,
        # termination_manager.py:149-150)
        terms = list(cm._terms.values())
        if len(terms) == 1 and set(terms[0].metrics) == {"error_vel_xy", "error_vel_yaw"}:
            # one launch: count, reward means, command metric means, termination counts (layout of the fused launch's log)
            if getattr(self, "_post_reset_log", None) is None:
                rows = 1 + len(rm.active_terms) + 2 + len(tm.active_terms)
                self._post_reset_log = torch.zeros(rows, dtype=torch.float32, device=self.device)
            self._next_reset_log()
            L.check(L.lib.mdp_reset_log(
                self.num_envs, mask.data_ptr(), rm._episode_sums_block.data_ptr(), len(rm.active_terms),
                terms[0].metrics["error_vel_xy"].data_ptr(), terms[0].metrics["error_vel_yaw"].data_ptr(),
                tm._term_dones_block.data_ptr(), len(tm.active_terms), float(self.max_episode_length_s),
                self._post_reset_log.data_ptr(), L.stream(self.device)))
            self._publish_reset_log()
            log = self.extras["log"]
            num_reset = self._post_reset_log
        else:
            m = mask.to(torch.float32)
            num_reset = m.sum()
            cnt = num_reset.clamp(min=1.0)
            log = {}
            sums = (rm._episode_sums_block[: len(rm.active_terms)] * m).sum(dim=1) / cnt / self.max_episode_length_s
            for i, name in enumerate(rm.active_terms):
                log["Episode_Reward/" + name] = sums[i]
            for term_name, term in cm._terms.items():
                for k, v in term.metrics.items():
                    log[f"Metrics/{term_name}/{k}"] = (v * m).sum() / cnt
            done_counts = (tm._term_dones_block[: len(tm.active_terms)] & mask).sum(dim=1)
            for i, name in enumerate(tm.active_terms):
                log["Episode_Termination/" + name] = done_counts[i]
        log.update(self.curriculum_manager.reset(mask))
        self.extras["log"] = log
        L.check(L.lib.mdp_masked_fill(arr, count, mask.data_ptr(), self.num_envs, L.stream(self.device)))
        if "reset" in self.event_manager.available_modes:                      # manager_based_rl_env.py:357-359
            self.event_manager.apply(mode="reset", env_ids=mask,
                                     global_env_step_count=self._sim_step_counter // self.cfg.decimation)
        self.event_manager.reset(mask)
        for term in self._reset_cmd_terms:
            term._launch(0.0, mask)           # CommandTerm.reset: resample (+ counter) for the flagged envs
        # manager_based_rl_env.py:221: write_data_to_sim() in the steps that reset at least one env -- the condition
        # stays on the device (the launch is empty when num_reset == 0)
        self.scene.write_data_to_sim(gate=num_reset)

    def _reset_idx(self, env_ids):
        """manager_based_rl_env.py:346-391 without events / curriculum (out of scope)."""
        self.curriculum_manager.compute(env_ids=env_ids)
        self.scene.reset(env_ids)
        if "reset" in self.event_manager.available_modes:
            self.event_manager.apply(mode="reset", env_ids=env_ids,
                                     global_env_step_count=self._sim_step_counter // self.cfg.decimation)
        self.extras["log"] = dict()
        for manager in (self.observation_manager, self.action_manager, self.reward_manager, self.curriculum_manager,
                        self.command_manager, self.event_manager, self.termination_manager):
            self.extras["log"].update(manager.reset(env_ids))
        self.episode_length_buf[env_ids] = 0
