"""the reference's process model as ONE process.

main.py runs the tracking frontend (main.py:325-446) and the backend (run_backend, main.py:73-163; relocalization,
main.py:28-71) as two processes that share keyframes through CUDA-IPC buffers, manager locks and 10 ms polling.
`SlamSystem` is the same state machine (Mode.INIT / TRACKING / RELOC, the same calls in the same order) in one
process.  `backend="inline"` (default) has the semantics of the reference's `single_thread: True` evaluation configs
(the backend task of a keyframe is finished before the next frame is tracked, main.py:391-395) and is deterministic;
`backend="thread"` runs the backend on its own host thread and HIP stream beside the tracking loop, as the reference's
second process does: the symmetric edge inference (most of a backend task) overlaps tracking, while everything that
reads or writes keyframe pointmaps / poses (the tracking step on one side; global GN + TSDF hook on the other) runs in
critical sections that hand the data over between the two streams with events instead of locks + polling + IPC copies.
Two more things the split made impossible:

* **frame groups** (`frame_group` = B > 1): the encoder runs B frames ahead on its own stream and the two-view
  forward of the next B frames against the current keyframe is ONE batch call
  (mast3r_utils.mast3r_asymmetric_inference_group).  Matching and tracking stay strictly per frame, in order.  The
  group decode speculates that the keyframe stays: when a frame of the group becomes a keyframe (or tracking falls
  into RELOC) the rest of the group is decoded again against the new keyframe.  Rows of a batch are computed exactly as
  a batch of one, so the trajectory is bit-identical to B = 1 (tests/test_slam_system_gpu.py).
* the keyframe store is whatever the caller passes: `KeyframeStore` (unbounded) or the `SharedKeyframes` mirror.

Retrieval (retrieval_database.py, ASMK) is not available offline; `retriever` is any object with the reference's
`update(frame, add_after_query, k, min_thresh) -> [keyframe ids]`.  The default stand-in proposes no loop-closure edges
and offers the most recent keyframes as relocalisation candidates."""
import contextlib
import os
import queue
import threading
import time

import torch

from lietorch_hip import Sim3
from mast3r_slam import mast3r_utils as mu
from mast3r_slam.config import config
from mast3r_slam.frame import KeyframeStore, Mode
from mast3r_slam.global_opt import FactorGraph
from mast3r_slam.tracker import FrameTracker


class RecentKeyframes:
    """Stand-in for RetrievalDatabase.update: no loop-closure proposals for the graph (add_after_query=True), the
    last `k` keyframes as relocalisation candidates (add_after_query=False)."""

    def __init__(self, keyframes, k=1):
        self.keyframes, self.k = keyframes, k

    def update(self, frame, add_after_query=True, k=3, min_thresh=0.0):
        if add_after_query:
            return []
        n = len(self.keyframes)
        return list(range(max(0, n - self.k), n))[::-1]


class _BackendThread(threading.Thread):
    """One stage of run_backend (main.py:73-163) as a host thread with its own stream.  Tasks are (keyframe index, event
    the stage's stream waits for, payload of the previous stage); `fn(idx, payload)` does the stage's work and its return
    value travels to `next_stage` behind an event recorded on this stage's stream."""

    def __init__(self, system, fn, next_stage=None, priority=0):
        super().__init__(daemon=True)
        self.system, self.fn, self.next_stage, self.priority = system, fn, next_stage, int(priority)
        self.q, self.error = queue.Queue(), None
        self.start()

    def run(self):
        dev = self.system.device
        torch.cuda.set_device(dev)
        with torch.cuda.stream(torch.cuda.Stream(device=dev, priority=self.priority)):
            while True:
                task = self.q.get()
                try:
                    if task is None:
                        return
                    if self.error is None:
                        idx, ev, payload = task
                        stream = torch.cuda.current_stream(dev)
                        stream.wait_event(ev)              # what the task reads was produced on another stream
                        out = self.fn(idx, payload)
                        if self.next_stage is not None:
                            done = torch.cuda.Event()
                            done.record(stream)
                            self.next_stage.q.put((idx, done, out))
                except Exception as e:   # surfaced by drain()
                    self.error = e
                finally:
                    self.q.task_done()

    def drain(self):
        self.q.join()                    # every task of this stage has been handed on before it counts as done
        if self.error is not None:
            raise self.error
        if self.next_stage is not None:
            self.next_stage.drain()

    def stop(self):
        st = self
        while st is not None:
            st.q.put(None)
            st.join(timeout=5.0)
            st = st.next_stage


class SlamSystem:
    def __init__(self, model, device, K=None, keyframes=None, retriever=None, frame_group=1, tsdf_global_cfg=None,
                 encoder_group=None,
                 backend="inline", tsdf_refine_cfg=None, quality_service=None, shard_edges=False, decode_ahead=0,
                 shard_channel=None, pipeline=False, pipeline_depth=1, backend_priority=0, encoder_priority=0,
                 backend_stages=3, solve_priority=0):
        """`shard_channel` (mast3r_slam/shard.py): this process is the DRIVER rank of a session whose backend is sharded
        over the ranks of the channel's group - keyframe-pair inference + matching, the global GN (one all-reduce per
        iteration) and the global TSDF's voxels; the other ranks run BackendShard.serve()."""
        self.model, self.device, self.K = model, torch.device(device), K
        self.keyframes = KeyframeStore() if keyframes is None else keyframes
        self.tracker = FrameTracker(model, self.keyframes, device)
        self.tracker.quality_service = quality_service          # main.py:246
        self.shard_channel = shard_channel
        self.backend_priority = int(backend_priority)      # HIP stream priority of the backend thread's stream (-1 = high)
        self.pipeline = bool(pipeline)
        self.pipeline_depth = min(3, max(1, int(pipeline_depth)))     # FrameTracker keeps 4 solver states
        self.factor_graph = FactorGraph(model, self.keyframes, K, device, shard_edges=shard_edges or shard_channel is not None,
                                        channel=shard_channel)
        self.retriever = RecentKeyframes(self.keyframes) if retriever is None else retriever
        self.tsdf_manager = None
        if tsdf_global_cfg is not None and tsdf_global_cfg.get("enabled", False):   # main.py:78-88
            from mast3r_slam.tsdf import TSDFGlobalManager

            self.tsdf_manager = TSDFGlobalManager(self.keyframes, tsdf_global_cfg, config.get("use_calib", False), device,
                                                  channel=shard_channel)
            self.tsdf_manager.start()
        # the camera-side half of the dual TSDF (main.py:253-287): local block refinement of keyframes that left the
        # sliding window; scheduled and processed by the backend task (synchronous form of the refiner thread)
        self.tsdf_refiner = None
        if tsdf_refine_cfg is not None and tsdf_refine_cfg.get("enabled", False):
            from mast3r_slam.tsdf_refine import TSDFRefiner

            self.tsdf_refiner = TSDFRefiner(tsdf_refine_cfg, self.keyframes, quality_service, device)
            self.tsdf_refiner.edit_section = lambda: self._critical("refine")
            self.tsdf_refiner.start()
        assert backend in ("inline", "thread")
        self._lock = threading.RLock()
        self._bprof = bool(os.environ.get("MSLAM_BACKEND_PROFILE"))
        self._hand = {"main": None, "backend": None, "refine": None}   # event at the end of each party's last critical section
        self._commits = []                              # (solve job, event): optimised poses waiting to be written back
        self._backend_done = None                       # event behind the last backend task (threaded backend)
        # backend="thread", staged (default 3): the graph stage of keyframe k+1 (retrieval + symmetric inference +
        # matching: throughput-bound network launches) runs beside the solve stage of keyframe k (global GN: a dependent
        # chain of short kernels) and the fusion / refinement stage of keyframe k-1 (TSDF hook, local refinement: short
        # launches with host reads in between), each on a thread / stream of its own - the keyframe task's chain (64 ms
        # per keyframe at 60-120 keyframes, of which 27 ms network) was what bound the loop.
        # Every solve still sees exactly the edges of the tasks up to its own (the edge count travels with the task).
        # A sharded session keeps ONE backend thread: its collectives must be issued in one order on every rank.
        self._worker = None
        self._graph_done = None
        if backend == "thread":
            if int(backend_stages) >= 2 and shard_channel is None:
                post = None
                if int(backend_stages) >= 3:     # TSDF fusion + local refinement (host reads) behind the solve, on their own
                    post = _BackendThread(self, lambda idx, payload: self._backend_post(idx, payload), priority=solve_priority)
                solve = _BackendThread(self, lambda idx, n_edges: self._backend_solve(idx, n_edges, hand_on=post is not None),
                                       next_stage=post, priority=solve_priority)
                self._worker = _BackendThread(self, lambda idx, _: self._backend_graph(idx), next_stage=solve,
                                              priority=self.backend_priority)
            else:
                self._worker = _BackendThread(self, lambda idx, _: self._backend(idx), priority=self.backend_priority)
        self.mode = Mode.INIT
        self.last_T = None
        self.frame_group = max(1, int(frame_group))
        # the look-ahead encoder does not depend on any decision of the loop (every frame is encoded exactly once), so
        # its batch may be larger than the speculative decode group: nothing is ever encoded in vain
        self.encoder_group = self.frame_group if encoder_group is None else max(1, int(encoder_group))
        self.enc_stream = torch.cuda.Stream(device=self.device, priority=int(encoder_priority)) if self.frame_group > 1 else None
        self._enc_hi = 0
        self._kf_value, self._kf_slope = None, None      # keyframe-rule value of the last tracked frame, its decay per frame
        self._n_pending = 0                              # frames begun whose verdict is not read yet (pipelined run)
        # decode_ahead = k > 0: the NEXT group's pair decode is issued on a stream of its own as soon as at most k
        # already decoded frames are left in front of the current one, so that it runs beside the per-frame matching /
        # tracking of the current group instead of in front of the next group's first frame (more rows are decoded
        # in vain when the keyframe changes)
        self.decode_ahead = max(0, int(decode_ahead)) if self.frame_group > 1 else 0
        self.dec_stream = torch.cuda.Stream(device=self.device) if self.decode_ahead > 0 else None
        self.stats = dict(frames=0, keyframes=0, group_calls=0, decoded_rows=0, void_rows=0, relocalised=0)

    # ------------------------------------------------------------------ frontend (main.py:325-446)
    def run(self, frames, start=0, stop=None, release=False):
        """Track frames[start:stop] (Frame objects, frame.create_frame) in order; returns the per-frame results of
        step().  A sequence may be fed in several calls over the SAME list (the look-ahead state carries over).
        `release` drops the list's reference to a frame once it has been tracked (keyframes live on in the store), so
        that a long sequence does not keep every pointmap alive.

        `pipeline` (off by default, see below): in TRACKING mode the matching + pose solve of the next `pipeline_depth`
        frames are enqueued BEFORE frame f's verdict is read - on the premise "f is tracked and the keyframe stays" (true
        for ~88 % of the frames) - so the one host wait per frame no longer leaves the tracking stream without work.  When the premise fails (new keyframe, tracking
        lost, a solve that needed more than its first chunk of iterations) f+1 is rolled back (nothing of it has
        reached the keyframe store: FrameTracker keeps the fused keyframe in a shadow copy until the verdict is in) and
        begun again from f's real outcome.  Results are bit-identical to the frame-at-a-time loop.

        MEASURED (1 000-frame stream, same box, DESIGN.md "Round 3: the tracking loop"): the frontend's host work drops
        from 5.7 to 2.0-2.9 ms per frame, but the JOB gets slower (138 -> 133 frames/s at depth 1, 116-126 at depth 2 or with
        one hand-over section per iteration): a tracking stream that never runs dry takes the device away from the
        backend stream, the frontend finishes early and the backend drains its backlog alone - a single dependent
        chain of short kernels with a launch bubble behind each.  The frame-at-a-time loop interleaves the two by
        itself (the device works on the backend while the host reads a verdict), so it stays the default."""
        out = []
        stop = len(frames) if stop is None else min(stop, len(frames))
        self._enc_hi = max(self._enc_hi, start)
        if not self.pipeline:
            for i in range(start, stop):
                if self.frame_group > 1:
                    self._look_ahead(frames, i, stop)
                out.append(self.step(frames[i]))
                if release:
                    frames[i] = None
            return out
        i, pend = start, []                         # pend = [(index, handle)]: begun, verdict not read yet (oldest first)
        while i < stop or pend:
            # run ahead: up to `pipeline_depth` frames begun behind the one whose verdict is read next (the host then
            # has a frame's worth of enqueue time in hand when the device finishes a frame)
            while i < stop and self.mode == Mode.TRACKING and len(pend) <= self.pipeline_depth:
                # speculation that is rarely wrong: the value the keyframe rule compares with its threshold decays
                # steadily (see _speculative_window); when it says that a frame still in flight will replace the
                # keyframe, nothing more is begun (or decoded) behind it until that verdict is in
                self._n_pending = len(pend)
                if pend and not self._premise_holds(len(pend)):
                    break
                if self.frame_group > 1:
                    self._look_ahead(frames, i, stop)
                pend.append((i, self._begin(frames[i])))
                i += 1
            self._n_pending = 0
            if pend:
                k, h = pend.pop(0)
                with self._critical("main"):
                    self.tracker.track_resolve(h)
                    clean = h.kind == "ok" and not h.new_kf and not h.replayed
                    if pend and not clean:              # the premise of everything begun behind it failed: undo, newest first
                        for _, hh in reversed(pend):
                            self.tracker.rollback(hh)
                        self.stats["replayed_frames"] = self.stats.get("replayed_frames", 0) + len(pend)
                        i = pend[0][0]
                        pend = []
                    res, add_new_kf = self._end(h)
                if add_new_kf:
                    self._queue_backend(len(self.keyframes) - 1)
                if not pend:
                    self.last_T = h.frame.T_WC
                out.append(res)
                if release:
                    frames[k] = None
            elif i < stop and self.mode != Mode.TRACKING:      # INIT / RELOC: frame at a time
                if self.frame_group > 1:
                    self._look_ahead(frames, i, stop)
                out.append(self.step(frames[i]))
                if release:
                    frames[i] = None
                i += 1
        return out

    def _begin(self, frame):
        """First half of step() for a frame in TRACKING mode: everything enqueued, verdict not read."""
        self._wait_encoded(frame)
        self._wait_decoded(frame)
        if self.last_T is not None:
            frame.T_WC = Sim3(self.last_T.data.clone())
        with self._critical("main"):
            self._apply_commits()
            h = self.tracker.track_begin(frame)
        self.last_T = frame.T_WC                    # optimistic: the next frame starts from this one's solved pose
        return h

    def _end(self, h):
        """Second half (inside the hand-over section): the verdict's consequences -> (step()'s result dict, new_kf)."""
        frame = h.frame
        self.stats["frames"] += 1
        add_new_kf, _, try_reloc = self.tracker.track_finish(h)
        if try_reloc:
            self.mode = Mode.RELOC
        self._note_keyframe_rule(add_new_kf or try_reloc)
        if add_new_kf:
            self.keyframes.append(frame)
            self.stats["keyframes"] += 1
        return (dict(mode=Mode.TRACKING, new_kf=bool(add_new_kf), try_reloc=bool(try_reloc), pose=frame.T_WC.data.clone()),
                bool(add_new_kf))

    def finish(self):
        """End of the sequence (main.py:449-560): drain the backend, then the local refiner's final pass over the
        keyframes still inside its window."""
        self.drain()
        if self.tsdf_refiner is not None and len(self.keyframes) > 0:
            with self._critical("main"):
                self.tsdf_refiner.schedule_final_pass(len(self.keyframes) - 1)
                self.tsdf_refiner.process_queue()

    def step(self, frame):
        """One iteration of the main loop for an already created frame -> dict(mode, new_kf, try_reloc, pose) with
        `pose` = the frame's T_WC data as tracked (before any later backend update of a keyframe copy)."""
        self._wait_encoded(frame)
        self._wait_decoded(frame)
        if self.last_T is not None:                     # "last camera pose for the frame" (main.py:351-356)
            frame.T_WC = Sim3(self.last_T.data.clone())
        self.stats["frames"] += 1
        add_new_kf = try_reloc = False
        mode = self.mode
        if mode == Mode.RELOC and self._worker is not None:
            self._worker.drain()                        # relocalisation edits the factor graph: the backend must be idle
            self._adopt_backend_tensors()
        with self._critical("main"):
            self._apply_commits(wait=(mode == Mode.RELOC))
            if mode == Mode.INIT:                       # main.py:359-367
                X_init, C_init = mu.mast3r_inference_mono(self.model, frame)
                frame.update_pointmap(X_init, C_init)
                add_new_kf = True
                self.mode = Mode.TRACKING
            elif mode == Mode.TRACKING:                 # main.py:369-373
                h = self.tracker.track_begin(frame)
            elif mode == Mode.RELOC:                    # main.py:375-385
                self.tracker._shadow = None             # whatever comes next is tracked against the store's keyframe
                X, C = mu.mast3r_inference_mono(self.model, frame)
                frame.update_pointmap(X, C)
                if self._relocalization(frame):
                    self.mode = Mode.TRACKING
                    self.stats["relocalised"] += 1
            else:
                raise Exception("Invalid mode")
            if mode != Mode.TRACKING:
                self.last_T = frame.T_WC
                if add_new_kf:                          # main.py:387-395
                    self.keyframes.append(frame)
                    self.stats["keyframes"] += 1
        if mode == Mode.TRACKING:
            # the one host wait of a tracked frame happens OUTSIDE the hand-over section: the backend's solve stage can copy
            # keyframe data out (or write poses back) while this thread sleeps on the verdict
            self.tracker.track_resolve(h)
            with self._critical("main"):
                add_new_kf, _, try_reloc = self.tracker.track_finish(h)
                if try_reloc:
                    self.mode = Mode.RELOC
                self._note_keyframe_rule(add_new_kf or try_reloc)
                self.last_T = frame.T_WC
                if add_new_kf:                          # main.py:387-395
                    self.keyframes.append(frame)
                    self.stats["keyframes"] += 1
        if add_new_kf:
            self._queue_backend(len(self.keyframes) - 1)
        return dict(mode=mode, new_kf=bool(add_new_kf), try_reloc=bool(try_reloc), pose=frame.T_WC.data.clone())

    def drain(self):
        """Wait until every queued backend task has been issued and its poses are written back (backend="thread")."""
        if self._worker is not None:
            self._worker.drain()
            main = torch.cuda.current_stream(self.device)
            for ev in (self._backend_done, self._graph_done, self._hand["backend"], self._hand["refine"]):
                if ev is not None:           # the stages' last launches (the refiner's edits, the edge lists) lie
                    main.wait_event(ev)      # behind their last hand-over sections
            with self._critical("main"):
                self._apply_commits(wait=True)

    def _adopt_backend_tensors(self):
        """Relocalisation (main.py:28-71) edits and solves the factor graph on the TRACKING stream, but its tensors were
        allocated and last written on the backend stream: the tracking stream first waits for everything the (now idle)
        backend thread has enqueued, and the caching allocator is told that these blocks are in use on this stream too."""
        main = torch.cuda.current_stream(self.device)
        for ev in (self._backend_done, self._graph_done):
            if ev is not None:
                main.wait_event(ev)
        fg = self.factor_graph
        for t in (fg.ii, fg.jj, fg.idx_ii2jj, fg.idx_jj2ii, fg.valid_match_j, fg.valid_match_i, fg.Q_ii2jj, fg.Q_jj2ii):
            if t.is_cuda and t.numel():
                t.record_stream(main)

    def _apply_commits(self, wait=False):
        """Threaded backend: the global GN runs on copies, outside the hand-over lock; its poses are written into the
        store HERE, by the tracking side on its own stream, once the solve has finished on the device (the reference's
        backend process writes them whenever it is done, main.py:145-164 - the frontend never waits for a solve).
        Always called inside a hand-over section (the backend calls it too, in front of the TSDF pose optimiser)."""
        while self._commits:
            job, ev = self._commits[0]
            if not wait and not ev.query():
                return
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for t in job.values():      # allocated on the backend's stream, read here on this one: without the note the
                if torch.is_tensor(t) and t.is_cuda:   # caching allocator hands the block back to the backend the moment
                    t.record_stream(cur)               # the job is dropped, while this stream's copy is still queued
            self.factor_graph.commit_solve(job)
            self._commits.pop(0)

    def shutdown(self):
        if self._worker is not None:
            self._worker.drain()
            self._worker.stop()
            self._worker = None
        if self.tsdf_manager is not None:
            self.tsdf_manager.shutdown()

    @contextlib.contextmanager
    def _critical(self, me):
        """Section that reads or writes keyframe pointmaps / poses.  Inline backend: nothing to do.  Threaded backend:
        one party at a time (host lock; parties: "main" = tracking, "backend" = the solve stage, "refine" = the fusion /
        refinement stage), and the entering party's stream first waits for the events the OTHER parties recorded when they
        left their last sections, so data written on one stream is visible to the others."""
        if self._worker is None:
            yield
            return
        with self._lock:
            stream = torch.cuda.current_stream(self.device)
            for other, ev in self._hand.items():
                if other != me and ev is not None:
                    stream.wait_event(ev)
            try:
                yield
            finally:
                ev = torch.cuda.Event()
                ev.record(stream)
                self._hand[me] = ev

    def _queue_backend(self, idx):
        if self._worker is None:
            self._backend(idx)
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._worker.q.put((idx, ev, None))

    # ------------------------------------------------------------------ frame groups
    def _note_keyframe_rule(self, changed):
        """Bookkeeping for the size of the next speculative decode: the value the keyframe rule compares with its
        threshold (tracker.py:170-177) falls steadily while the camera moves away from the keyframe; its decay per frame
        says how many more frames the keyframe is likely to last."""
        v = self.tracker.last_kf_value
        if changed or v is None:
            self._kf_value = None
            return
        if self._kf_value is not None:
            d = max(self._kf_value - v, 0.0)
            self._kf_slope = d if self._kf_slope is None else 0.5 * self._kf_slope + 0.5 * d
        self._kf_value = v

    def _speculative_window(self, B, already=0):
        """Frames worth decoding against the current keyframe in one call: all B right behind a keyframe change, fewer
        when the keyframe rule is about to fire (rows decoded against a keyframe that is replaced are wasted; results do
        not depend on the grouping).  `already` = frames in front of the call that are decoded but not tracked yet
        (look-ahead calls): they use up part of what the keyframe is expected to last."""
        if self._kf_value is None or not self._kf_slope or self._kf_slope <= 0.0:
            return B if already == 0 else 0      # nothing known yet: no speculation beyond the current group
        left = (self._kf_value - config["tracking"]["match_frac_thresh"]) / self._kf_slope
        n = min(B, int(left) - already)
        return max(1, n) if already == 0 else max(0, n)

    def _premise_holds(self, in_flight):
        """Are the `in_flight` frames whose verdict is still out all predicted to leave the keyframe in place?  (The same
        prediction that sizes the group decode; unknown decay = right behind a keyframe change: no.)"""
        if self._kf_value is None or not self._kf_slope or self._kf_slope <= 0.0:
            return False
        left = (self._kf_value - config["tracking"]["match_frac_thresh"]) / self._kf_slope
        return in_flight <= int(left)

    def _wait_encoded(self, frame, stream=None, keep=False):
        ev = getattr(frame, "enc_event", None)
        if ev is not None:
            stream = torch.cuda.current_stream(self.device) if stream is None else stream
            stream.wait_event(ev)
            frame.feat.record_stream(stream)
            frame.pos.record_stream(stream)
            if not keep:
                frame.enc_event = None

    def _wait_decoded(self, frame):
        ev = getattr(frame, "dec_event", None)
        if ev is not None:
            main = torch.cuda.current_stream(self.device)
            main.wait_event(ev)
            stash = getattr(frame, "decoded", None)
            if stash is not None:
                for t in stash[1]:
                    t.record_stream(main)
            frame.dec_event = None

    def _look_ahead(self, frames, i, stop=None):
        B, n = self.frame_group, (len(frames) if stop is None else stop)
        EB = max(B, self.encoder_group)
        main = torch.cuda.current_stream(self.device)
        while self._enc_hi < min(n, i + EB + B):         # encoder: groups of EB, at least one decode group ahead of frame i
            grp = [frames[k] for k in range(self._enc_hi, min(n, self._enc_hi + EB))]
            self.enc_stream.wait_stream(main)            # the images were produced on the caller's stream
            with torch.cuda.stream(self.enc_stream):
                mu.encode_frames(self.model, grp)
                ev = torch.cuda.Event()
                ev.record()
            for f in grp:
                f.enc_event = ev
            self._enc_hi += len(grp)
        if self.mode != Mode.TRACKING:
            return
        keyframe = self.keyframes.last_keyframe()
        kid = int(keyframe.frame_id)

        def decoded(k):
            stash = getattr(frames[k], "decoded", None)
            return stash is not None and stash[0] == kid

        h = i - 1                                        # last frame of the run of frames decoded against this keyframe
        while h + 1 < n and decoded(h + 1):
            h += 1
        if h < i:
            lo = i                                       # the current frame itself is missing: it has to wait
        elif self.decode_ahead > 0 and h - i < self.decode_ahead and h + 1 < n:
            lo = h + 1                                   # few left: the next group, beside this one's tracking
        else:
            return
        # frames begun but not resolved yet (pipelined run) use up part of what the keyframe is expected to last, too
        window = [frames[k] for k in range(lo, min(n, lo + self._speculative_window(B, lo - i + self._n_pending)))]
        if not window and lo == i:
            window = [frames[i]]                         # the frame about to be tracked itself
        if not window:
            return
        for f in window:
            if getattr(f, "decoded", None) is not None:  # decoded against a keyframe that has been replaced since
                self.stats["void_rows"] += 1
        if self.dec_stream is None:
            for f in window:
                self._wait_encoded(f)
            mu.mast3r_asymmetric_inference_group(self.model, window, keyframe)
        else:
            self.dec_stream.wait_stream(main)            # the keyframe (and its features) exist on the tracking stream
            for f in window:
                self._wait_encoded(f, self.dec_stream, keep=True)
            with torch.cuda.stream(self.dec_stream):
                mu.mast3r_asymmetric_inference_group(self.model, window, keyframe)
                ev = torch.cuda.Event()
                ev.record()
            for f in window:
                f.dec_event = ev
        self.stats["group_calls"] += 1
        self.stats["decoded_rows"] += len(window)

    # ------------------------------------------------------------------ backend (main.py:73-163, 28-71)
    def _phase(self, name, t0):
        """MSLAM_BACKEND_PROFILE=1 (diagnostic): wall time per backend phase, each closed by a stream synchronise - what
        the keyframe task's dependent chain costs, phase by phase, beside a running frontend."""
        if not self._bprof:
            return t0
        torch.cuda.current_stream(self.device).synchronize()
        t1 = time.perf_counter()
        acc = self.stats.setdefault("backend_phase_s", {})
        acc[name] = acc.get(name, 0.0) + (t1 - t0)
        return t1

    def _backend(self, idx):
        """The body of run_backend's loop for the keyframe `idx` (graph construction, global GN, TSDF hook)."""
        self._backend_solve(idx, self._backend_graph(idx))

    def _backend_graph(self, idx):
        """First half of the keyframe task (main.py:118-143): retrieval, symmetric edge inference + matching, the new
        edges appended -> the number of edges the graph holds with this task's (the solve of this task uses those)."""
        t0 = time.perf_counter()
        kf_idx = []
        n_consec = 1
        for j in range(min(n_consec, idx)):
            kf_idx.append(idx - 1 - j)
        frame = self.keyframes[idx]
        retrieval_inds = self.retriever.update(frame, add_after_query=True, k=config["retrieval"]["k"],
                                               min_thresh=config["retrieval"]["min_thresh"])
        kf_idx += retrieval_inds
        kf_idx = set(kf_idx)
        kf_idx.discard(idx)
        kf_idx = list(kf_idx)
        frame_idx = [idx] * len(kf_idx)
        t0 = self._phase("retrieval", t0)
        if kf_idx:   # symmetric edge inference + matching: reads only the keyframes' (immutable) features
            self.factor_graph.add_factors(kf_idx, frame_idx, config["local_opt"]["min_match_frac"])
        self._phase("add_factors", t0)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(self.device))
        self._graph_done = done
        return self.factor_graph.n_edges

    def _backend_solve(self, idx, n_edges, hand_on=False):
        """Second half (main.py:145-163): global GN over the graph as it stood when this keyframe's edges had been added;
        then (or on the next stage's thread, `hand_on`) the TSDF hook and the local refinement (_backend_post)."""
        t0 = time.perf_counter()
        kind = "calib" if config["use_calib"] else "rays"
        if self._worker is None:     # inline: the reference's single_thread order, everything in sequence
            self._solve()
            t0 = self._phase("solve", t0)
            if self.tsdf_manager is not None:
                self.tsdf_manager.on_after_backend_solve(self.factor_graph)
            t0 = self._phase("tsdf", t0)
            self._refine(idx)
            self._phase("refine", t0)
            return None
        # threaded: the lock (and with it the tracking stream) is held only while keyframe data is copied out; the
        # solve and the fusions run on the copies; the poses are written back by the tracking side (_apply_commits)
        mgr = self.tsdf_manager
        with self._critical("backend"):
            t0 = self._phase("lock_wait", t0)
            job = self.factor_graph.prepare_solve(kind, n_edges=n_edges)
            if job is not None:
                self._chain_pending_poses(job)
            plan = mgr.plan(self.factor_graph) if mgr is not None else None
            t0 = self._phase("prepare+plan", t0)
        if job is not None:
            self.factor_graph.run_solve(job)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            with self._lock:
                self._commits.append((job, ev))
        self._phase("solve", t0)
        payload = (plan, job)
        if hand_on:
            return payload
        self._backend_post(idx, payload)
        return None

    def _backend_post(self, idx, payload):
        """The TSDF hook of the keyframe task (fusions at the poses its solve produced) and the local refinement
        (main.py:403-421).  With three backend stages this runs on a thread / stream of its own: both are chains of short
        launches with host reads in between, which would otherwise hold up the next solve."""
        t0 = time.perf_counter()
        plan, job = payload
        mgr = self.tsdf_manager
        cur = torch.cuda.current_stream(self.device)
        if plan is not None:
            for _, _, snap in plan["todo"]:        # copies taken on the solve stage's stream, read (and dropped) here
                for t in (snap or ()):
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(cur)
            if job is not None:      # the fusions use the poses THIS solve produced, as the inline order does
                job["pose_data"].record_stream(cur)
                mgr.retarget(plan, job["unique_kf_idx_host"], job["pose_data"])
            if not mgr.has_pose_refinement(plan):
                mgr.execute(plan)            # fusions only: on the copies, outside the lock
            else:
                # the TSDF pose optimiser reads and writes keyframe poses in the store: inside the hand-over section,
                # behind this solve's own write-back (rare configuration: the tracking stream waits for the solve here)
                with self._critical("refine"):
                    self._apply_commits(wait=True)
                    mgr.execute(plan)
        t0 = self._phase("tsdf", t0)
        if self.tsdf_refiner is not None:
            # the refiner works on keyframes that have LEFT the sliding window (the tracking side reads and replaces only
            # the newest one): no section around the whole pass - an empty one orders this stream behind the other parties'
            # last writes (poses), and the refiner's in-place edit of a keyframe is a section of its own (edit_section)
            with self._critical("refine"):
                pass
            self._refine(idx)
            t0 = self._phase("refine", t0)
        done = torch.cuda.Event()
        done.record(cur)
        self._backend_done = done

    def _chain_pending_poses(self, job):
        """A solve whose result is still waiting in `_commits` (the tracking side writes it into the store when it gets
        there) has not reached the store this job's poses were copied from: take its poses - they were produced on
        this (the backend's) stream, so stream order makes them visible - for every keyframe both solves hold.  Called
        under the hand-over lock.  Without it solve N+1 restarts from pre-N poses and later overwrites N's result."""
        cur = {int(k): r for r, k in enumerate(job["unique_kf_idx_host"].tolist())}
        for prev, _ in self._commits:
            pin = prev["pin"]
            ids = prev["unique_kf_idx_host"].tolist()
            src = [r for r in range(pin, len(ids)) if int(ids[r]) in cur]
            if not src:
                continue
            dst = [cur[int(ids[r])] for r in src]
            dev = job["pose_data"].device
            job["pose_data"][torch.tensor(dst, device=dev)] = prev["pose_data"][torch.tensor(src, device=dev)]

    def _refine(self, idx):
        if self.tsdf_refiner is not None:      # main.py:403-421, after the backend task of the keyframe
            self.tsdf_refiner.registry.tick()
            self.tsdf_refiner.maybe_schedule_sliding_window(idx)
            self.stats["refine_blocks"] = self.stats.get("refine_blocks", 0) + self.tsdf_refiner.process_queue()

    def _solve(self):
        if config["use_calib"]:
            self.factor_graph.solve_GN_calib()
        else:
            self.factor_graph.solve_GN_rays()

    def _relocalization(self, frame):
        """main.py:28-71."""
        kf_idx = list(self.retriever.update(frame, add_after_query=False, k=config["retrieval"]["k"],
                                            min_thresh=config["retrieval"]["min_thresh"]))
        success = False
        if kf_idx:
            self.keyframes.append(frame)
            n_kf = len(self.keyframes)
            frame_idx = [n_kf - 1] * len(kf_idx)
            if self.factor_graph.add_factors(frame_idx, kf_idx, config["reloc"]["min_match_frac"],
                                             is_reloc=config["reloc"]["strict"]):
                self.retriever.update(frame, add_after_query=True, k=config["retrieval"]["k"],
                                      min_thresh=config["retrieval"]["min_thresh"])
                success = True
                kf = self.keyframes[n_kf - 1]
                kf.T_WC = Sim3(self.keyframes[kf_idx[0]].T_WC.data.clone())
                self.keyframes[n_kf - 1] = kf
                self.stats["keyframes"] += 1
            else:
                self.keyframes.pop_last()
        if success:
            self._solve()
        return success
