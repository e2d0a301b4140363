"""``render(...)`` of fixed shapes captured in a HIP graph.

With the drop-in renderer a small-batch ``render()`` call is no longer bound by the GPU but by the ~380 kernel launches of
the PyTorch plane producer in front of it (run.py:221 -> models/generator.py:407-503: mapping network, StyleGAN2 synthesis,
texture mapper): one image x 128 x 128 x (64 + 64) keeps the GPU busy for 3.8 ms of a 5.5 ms call (profiles/r6/e2e/
split_render_b1_hip.json).  A HIP graph replays the whole call - producer, texel hand-off, decoder pack, the two noise draws,
ray set-up, the fused render kernel - as one submission:

    render = nfi_render.make_render(args, dataset_config, strict_near_far=False)
    graphed = GraphedRender(render, model, 128, 128, cam, focal, None, bbox, ws, 64)     # warm-up + capture
    rgb, depth, mask, normals, extra, _ = graphed(cam2, focal2, None, bbox2, ws2)       # copy-in + replay

Inference only (no gradient).  The model's parameters are read where they live: in-place updates (an optimiser step, the EMA
of run.py:365-377) are seen by the next replay, re-assigned parameter tensors are not.  The outputs are the graph's own tensors: they are overwritten by the next call, clone what
has to outlive it.  Every replay draws fresh noise (PyTorch registers its Philox state with the graph), exactly like
consecutive eager calls.  ``strict_near_far`` has to be False: the reference's failure for a batch without a hit is a host
read-back (lib/nerf_utils.py:258), and a captured stream cannot be waited on; such a batch renders as background.
"""
import torch


class GraphedRender:
    def __init__(self, render_fn, target_model, height, width, tform_cam2world, focal_length, center, bbox, model_input,
                 depth_samples_per_ray, warmup=3, **render_kw):
        if not tform_cam2world.is_cuda:
            raise ValueError('GraphedRender: CUDA / HIP tensors only')
        for k in ('extra_model_outputs',):
            if render_kw.get(k):
                raise ValueError('GraphedRender: %s is not supported (inference maps only)' % k)
        opts = getattr(render_fn, 'options', None)
        if opts is not None and opts.strict_near_far is not False:
            raise ValueError('GraphedRender: bind the render function with strict_near_far=False (make_render(args, '
                             'dataset_config, strict_near_far=False)): the strict check is a host read-back of the hit '
                             'count, which a captured stream cannot serve')
        self._args = (render_fn, target_model, int(height), int(width), int(depth_samples_per_ray))
        self._kw = dict(render_kw)
        clone = (lambda t: None if t is None else t.detach().clone())
        self._static = [clone(tform_cam2world), clone(focal_length), clone(center), clone(bbox), clone(model_input)]
        dev = tform_cam2world.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):          # MIOpen picks its solvers, the allocator and the LDS attributes settle
                self._call()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        try:
            with torch.no_grad(), torch.cuda.graph(self.graph):
                self._out = self._call()
        except RuntimeError as e:
            raise RuntimeError('GraphedRender: the call could not be captured - bind the render function with '
                               'strict_near_far=False (a host read-back cannot be captured) and keep extra model outputs '
                               'out of it: %s' % (e,)) from e

    def _call(self):
        render_fn, model, h, w, s = self._args
        cam, focal, center, bbox, inp = self._static
        return render_fn(model, h, w, cam, focal, center, bbox, inp, s, **self._kw)

    def __call__(self, tform_cam2world, focal_length, center, bbox, model_input):
        for dst, src, name in zip(self._static, (tform_cam2world, focal_length, center, bbox, model_input),
                                  ('tform_cam2world', 'focal_length', 'center', 'bbox', 'model_input')):
            if (dst is None) != (src is None):
                raise ValueError('GraphedRender: %s was %s at capture' % (name, 'None' if dst is None else 'a tensor'))
            if dst is not None:
                if dst.shape != src.shape:
                    raise ValueError('GraphedRender: %s has shape %s, captured with %s' % (name, tuple(src.shape), tuple(dst.shape)))
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self._out
