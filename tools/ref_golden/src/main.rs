// K1 of BASELINE.json: the README density (10-dim iid N(3,1)), 4 chains, DiagNutsSettings::default(), x0 = zeros,
// seeded exactly like the reference's own Sampler (src/sampler.rs:1105-1106: seed_from_u64(seed), stream chain+1;
// the chain's generator is then drawn from it inside new_chain, src/sampler.rs:761).
// Output: {"seed":0,"num_tune":400,"num_draws":1000,"chains":[{"chain":c,"draws":[[..10..],..],"num_steps":[..],
// "step_size":[..],"diverging":[..]}]} with every f64 also given as its bit pattern ("draws_bits") so that the
// comparison can be bit-exact where the arithmetic contract allows and 1e-9 otherwise.
use std::collections::HashMap;

use nuts_rs::{Chain, CpuLogpFunc, CpuMath, CpuMathError, DiagNutsSettings, HasDims, LogpError, Settings};
use rand::{SeedableRng, rngs::ChaCha8Rng};   // as the reference imports it (src/sampler.rs:6)
use thiserror::Error;

#[derive(Debug)]
struct PosteriorDensity {}

#[derive(Debug, Error)]
enum PosteriorLogpError {}
impl LogpError for PosteriorLogpError {
    fn is_recoverable(&self) -> bool {
        false
    }
}

impl HasDims for PosteriorDensity {
    fn dim_sizes(&self) -> HashMap<String, u64> {
        vec![("unconstrained_parameter".to_string(), 10u64)].into_iter().collect()
    }
}

impl CpuLogpFunc for PosteriorDensity {
    type LogpError = PosteriorLogpError;
    type ExpandedVector = Vec<f64>;
    type FlowParameters = ();

    fn dim(&self) -> usize {
        10
    }

    fn logp(&mut self, position: &[f64], grad: &mut [f64]) -> Result<f64, Self::LogpError> {
        let mu = 3f64;
        let logp = position
            .iter()
            .copied()
            .zip(grad.iter_mut())
            .map(|(x, grad)| {
                let diff = x - mu;
                *grad = -diff;
                -diff * diff / 2f64
            })
            .sum();
        Ok(logp)
    }

    fn expand_vector<R: rand::Rng + ?Sized>(&mut self, _rng: &mut R, position: &[f64]) -> Result<Vec<f64>, CpuMathError> {
        Ok(position.to_vec())
    }
}

fn main() {
    let mut settings = DiagNutsSettings::default();
    settings.seed = 0;
    settings.num_chains = 4;
    let total = settings.num_tune + settings.num_draws;
    let mut chains = Vec::new();
    for chain in 0..4u64 {
        let mut rng = ChaCha8Rng::seed_from_u64(settings.seed);
        rng.set_stream(chain + 1);
        let math = CpuMath::new(PosteriorDensity {});
        let mut sampler = settings.new_chain(chain, math, &mut rng);
        sampler.set_position(&vec![0f64; 10]).expect("init");
        let (mut draws, mut bits, mut steps, mut eps, mut div) = (vec![], vec![], vec![], vec![], vec![]);
        for _ in 0..total {
            let (draw, progress) = sampler.draw().expect("draw");
            bits.push(draw.iter().map(|v| format!("{:016x}", v.to_bits())).collect::<Vec<_>>());
            draws.push(draw.to_vec());
            steps.push(progress.num_steps);
            eps.push(progress.step_size);
            div.push(progress.diverging);
        }
        chains.push(serde_json::json!({"chain": chain, "draws": draws, "draws_bits": bits, "num_steps": steps,
                                       "step_size": eps, "diverging": div}));
    }
    println!("{}", serde_json::json!({"seed": settings.seed, "num_tune": settings.num_tune,
                                      "num_draws": settings.num_draws, "chains": chains}));
}
