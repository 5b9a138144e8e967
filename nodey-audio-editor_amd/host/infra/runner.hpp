// infra/runner.hpp — the ~100-line stand-in for the reference's Runner (src/infra/runner.cpp:11-154) that the
// tests drive the processors with: one Audio_stream per link created by the PRODUCER pin's generate_func (:40),
// output pins map to a SET of products (fan-out, :42-45), one fiber per node on one OS thread (:65-69), every
// exception captured per node into `exception` + State::Error instead of propagating (:75-136).
// The editor's Graph (node/pin/link ids, DAG checks, JSON) is out of scope; links are given by name here.
#pragma once
#include "fiber.hpp"
#include "processor.hpp"

#include <any>
#include <atomic>

namespace infra
{
	class Runner
	{
	  public:

		enum class State { Ready, Running, Finished, Error };  // include/infra/runner.hpp:25-31

		struct Processor_resource
		{
			std::shared_ptr<Processor> processor;
			std::map<std::string, std::shared_ptr<Processor::Product>> input_payloads;
			std::map<std::string, std::set<std::shared_ptr<Processor::Product>>> output_payloads;
			std::atomic<bool> stop_source{false};
			std::atomic<State> state{State::Ready};
			std::any exception;
			std::string error_text;  // what() of the captured exception, for test output
		};

		struct Link { Id_t from; std::string from_pin; Id_t to; std::string to_pin; };

		void add_node(Id_t id, std::shared_ptr<Processor> processor);
		void add_link(const Link& link);
		void set_node_data(Id_t id, std::shared_ptr<std::any> data) { node_data[id] = std::move(data); }
		// build the payload maps, run all fibers to completion on the calling thread; true if no node ended in Error
		bool run();
		void request_stop();  // sets every stop_source (Runner::~Runner, :53-63)

		const std::map<Id_t, std::shared_ptr<Processor_resource>>& get_processor_resources() const { return processor_resources; }
		size_t context_switches() const { return switches; }

	  private:

		std::map<Id_t, std::shared_ptr<Processor_resource>> processor_resources;
		std::vector<Link> links;
		std::map<Id_t, std::shared_ptr<std::any>> node_data;
		size_t switches = 0;
	};
}
